#!/bin/bash
# CPU-only AddressSanitizer + UBSan build of the HOST side of csrc/plan.hip (layout, parameter table, 2-D embedding
# offsets; device code is not instrumented: GPU sanitizers are not available on this pool).  The other objects are the
# product's.  Output: h-denseformer_amd/lib/libhdf_hip_san.so; run it with the sanitizer runtimes preloaded
# (tests/test_cpu_sanitized_host.py, tools/san_plan_walk.py).
set -e
cd "$(dirname "$0")/../h-denseformer_amd"
python build.py > /dev/null
mkdir -p build/san
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC $SAN -c csrc/plan.hip -o build/san/plan.o
# every product object except plan.o (the list follows build.py's SOURCES: a source added there is linked here too)
objs=$(python -c "import build; print(' '.join('build/' + s[:-4] + '.o' for s in build.SOURCES if s != 'plan.hip'))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -o lib/libhdf_hip_san.so $objs build/san/plan.o
echo built lib/libhdf_hip_san.so
