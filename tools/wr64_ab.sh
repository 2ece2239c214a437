#!/bin/bash
# conv_wr_kernel (weights in registers, hdf_op_conv3d_wr) against the plan's own routing (conv_ws2_kernel) for the Cin = 64
# launches at 64^3 (VERDICT r05 #1b), same box, interleaved.   usage: tools/wr64_ab.sh [reps]
cd "$(dirname "$0")/.."
REPS=${1:-30}
for shape in "64 64 64" "64 32 64" "64 128 64"; do
  set -- $shape
  for round in 1 2; do
    echo -n "wr : "; python tools/conv_micro.py --wr 1 --cin $1 --cout $2 --size $3 --reps $REPS 2>/dev/null | tail -1
    echo -n "ws2: "; python tools/conv_micro.py --wr 0 --cin $1 --cout $2 --size $3 --reps $REPS 2>/dev/null | tail -1
  done
done
