python -m pytest tests -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tfp -o tfp --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tfp -name "*kernel_stats.csv" | head -1)
grep -E "enc_tail|upsample_bwd" $f | cut -c1-200
find gpurun_out/tfp -name "*kernel_trace.csv" -delete
python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])"
