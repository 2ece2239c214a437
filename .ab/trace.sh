cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/st -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/st -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py $f 60 > gpurun_out/step_trace.txt
rm -f $f
