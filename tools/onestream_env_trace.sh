#!/bin/bash
# one-stream kernel traces (stats) of the default arrangement and of one with an environment setting
# usage: tools/onestream_env_trace.sh "HDF_NO_FUSED_APPLY=1"   -> gpurun_out/one_def, gpurun_out/one_env
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SET=$1
cd /tmp; export TMPDIR=/tmp; export HDF_NO_ASYNC_WGRAD=1 HDF_NO_BRANCH_OVERLAP=1
rm -rf $REPO/gpurun_out/one_def $REPO/gpurun_out/one_env
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/one_def -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
export $SET
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/one_env -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for d in one_def one_env; do
  f=$(ls -t $REPO/gpurun_out/$d/*/*kernel_stats.csv | head -1)
  cp $f $REPO/gpurun_out/${d}_kernel_stats.csv
done
