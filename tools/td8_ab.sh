#!/bin/bash
# Same-box interleaved A/B of the 8x8x8 tile of conv_ws2_kernel (TDP = 8, the 32 -> 32 layers at the top level) against the
# 4x8x8 tile: the product library vs a variant built with `python h-denseformer_amd/build.py --name libhdf_hip_td4
# -DHDF_NO_WS2_TD8`.  Per-launch times of the three forms the step runs (plain, input transform, statistics epilogue), then
# the whole step.   usage: tools/td8_ab.sh [micro reps] [step pairs]
cd "$(dirname "$0")/.."
REPS=${1:-20}
PAIRS=${2:-3}
VAR=${VAR:-h-denseformer_amd/lib/libhdf_hip_td4.so}
for form in "--xf 0" "--xf 1" "--bs 1"; do
  for round in 1 2; do
    echo -n "td8: "; python tools/conv_micro.py --cin 32 --cout 32 --size 128 --reps $REPS $form 2>/dev/null | tail -1
    echo -n "td4: "; HDF_LIB_PATH=$VAR python tools/conv_micro.py --cin 32 --cout 32 --size 128 --reps $REPS $form 2>/dev/null | tail -1
  done
done
bash tools/step_ab.sh $VAR $PAIRS
