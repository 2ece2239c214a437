#!/bin/bash
# Kernel traces of the bench step with the persistent transformer kernels and with the launch chain (one stream and
# three streams), summarised per kernel family -> gpurun_out/chain_trace/
# usage: bash tools/chain_trace.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/chain_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in chain launches; do
  if [ $v = launches ]; then export HDF_NO_TF_CHAIN=1; else unset HDF_NO_TF_CHAIN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_3s -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  export HDF_NO_ASYNC_WGRAD=1 HDF_NO_BRANCH_OVERLAP=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_1s -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  unset HDF_NO_ASYNC_WGRAD HDF_NO_BRANCH_OVERLAP
done
unset HDF_NO_TF_CHAIN
cd $REPO
for v in chain launches; do
  f=$(ls -t $(find $OUT/${v}_1s -name "*kernel_stats.csv") | head -1)
  echo "== $v one stream"; grep -i "tf_chain\|tok_\|attn_\|tf_wgrad\|patch" $f | awk -F, '{print $1, $2, $3, $4}' | head -20
  t=$(ls -t $(find $OUT/${v}_3s -name "*kernel_trace.csv") | head -1)
  echo "== $v three streams"; python3 tools/branch_timeline.py $t | tail -12
done
