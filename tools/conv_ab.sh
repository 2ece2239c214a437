#!/bin/bash
# Same-box interleaved A/B of the stride-1 conv kernels: the product library against a variant build
# (python h-denseformer_amd/build.py --name libhdf_hip_nowr -DHDF_NO_CONV_WR).  Usage: tools/conv_ab.sh [reps] [xf list]
cd "$(dirname "$0")/.."
REPS=${1:-20}
XFS=${2:-0}
VAR=${VAR:-h-denseformer_amd/lib/libhdf_hip_nowr.so}
for shape in ${SHAPES:-"64 32 128" "32 32 128" "32 64 128" "64 64 64" "32 64 64" "64 32 64"}; do
  set -- $shape
  for xf in $XFS; do
    for round in 1 2; do
      echo -n "new: "; python tools/conv_micro.py --wr 1 --cin $1 --cout $2 --size $3 --reps $REPS --xf $xf 2>/dev/null | tail -1
      echo -n "old: "; HDF_LIB_PATH=$VAR python tools/conv_micro.py --cin $1 --cout $2 --size $3 --reps $REPS --xf $xf 2>/dev/null | tail -1
    done
  done
done
