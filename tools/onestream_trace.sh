#!/bin/bash
# one-stream kernel traces of the product and of a variant library (HDF_LIB_PATH): per-launch durations, kernel by kernel
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp; export HDF_NO_ASYNC_WGRAD=1 HDF_NO_BRANCH_OVERLAP=1
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/one_new -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
export HDF_LIB_PATH=$REPO/h-denseformer_amd/lib/libhdf_hip_prev.so
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/one_prev -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
