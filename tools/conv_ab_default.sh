#!/bin/bash
# Same-box interleaved A/B of the plan's DEFAULT conv routing: the product library against a variant build (VAR, default
# the libhdf_hip_nowr.so of a previous source state).  Usage: tools/conv_ab_default.sh [reps]
cd "$(dirname "$0")/.."
REPS=${1:-20}
VAR=${VAR:-h-denseformer_amd/lib/libhdf_hip_nowr.so}
for shape in "32 32 128 0" "32 32 128 1" "32 64 128 0" "64 32 128 1" "64 64 64 0" "64 64 64 1" "32 64 64 0" "64 32 64 0" "16 32 128 0"; do
  set -- $shape
  for round in 1 2; do
    echo -n "new: "; python tools/conv_micro.py --cin $1 --cout $2 --size $3 --reps $REPS --xf $4 2>/dev/null | tail -1
    echo -n "old: "; HDF_LIB_PATH=$VAR python tools/conv_micro.py --cin $1 --cout $2 --size $3 --reps $REPS --xf $4 2>/dev/null | tail -1
  done
done
