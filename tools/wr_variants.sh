#!/bin/bash
# Attribution builds of conv_wr.hip (each deletes one ingredient of the phase; results are wrong, timings are the point):
# libhdf_hip_wr_<tag>.so = the product objects with conv_wr.o rebuilt under -DWR_DBG_<tag>.  Then tools/wr_attrib.sh on the GPU box.
cd "$(dirname "$0")/../h-denseformer_amd"
python build.py > /dev/null
VARS=${VARS:-"NOSTAGE NOSTORE NOMFMA STAMPS"}
for v in $VARS; do
  mkdir -p build/wr_$v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=1000000 -DWR_DBG_$v -c csrc/conv_wr.hip -o build/wr_$v/conv_wr.o &
done
wait
for v in $VARS; do
  objs="build/conv_igemm.o build/conv_first.o build/unet_ops.o build/transformer.o build/transformer_fused.o build/transformer_chain.o build/loss.o build/plan.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhdf_hip_wr_$v.so $objs build/wr_$v/conv_wr.o
done
ls -la lib/
