# diagnostic: stride-2 weight-gradient kernel variants (run through gpurun from the repo root)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/wg2probe
mkdir -p $OUT
rm -f $OUT/time.txt
cd /tmp && export TMPDIR=/tmp
for v in "HDF_WGRAD_S2_OLD=1" "HDF_WGRAD_S2_SB=1" "HDF_WGRAD_S2_SB=2" "HDF_WGRAD_S2_SB=2 HDF_WGRAD_OLD_WGS=512"; do
  echo "== $v" >> $OUT/time.txt
  env $v python3 $REPO/tools/convt_micro.py --op wgrad2 --cin 64 --cout 32 --size 64 2>&1 | grep -v amdgpu.ids >> $OUT/time.txt
  env $v python3 $REPO/tools/convt_micro.py --op wgrad2 --cin 128 --cout 64 --size 32 2>&1 | grep -v amdgpu.ids >> $OUT/time.txt
  env $v python3 $REPO/tools/convt_micro.py --op wgrad2 --cin 256 --cout 128 --size 16 2>&1 | grep -v amdgpu.ids >> $OUT/time.txt
done
cat $OUT/time.txt
