#!/bin/bash
cd "$(dirname "$0")/.."
REPS=${1:-20}
VARS=${VARS:-"_wr_NOSTAGE _wr_NOSTORE _wr_NOMFMA _nowr"}
for shape in "64 32 128" "32 32 128"; do
  set -- $shape
  for round in 1 2; do
    for v in "" $VARS; do
      echo -n "libhdf_hip$v: "; HDF_LIB_PATH=h-denseformer_amd/lib/libhdf_hip$v.so python tools/conv_micro.py --cin $1 --cout $2 --size $3 --reps $REPS 2>&1 | tail -1
    done
  done
done
