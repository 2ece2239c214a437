cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tfp -o tfp --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tfp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('dense_','attn_','block_out','patch_embed','in_bwd','in_finalize','enc_tail','upsample','head_','bias_grad','maxpool','loss_','wgrad','adam','rocclr','pack','in_apply','instnorm')):
        print(f"{n[:70]:70s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
find gpurun_out/tfp -name "*kernel_trace.csv" -delete
