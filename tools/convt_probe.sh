# diagnostic: counters of the transposed-conv forward / gather-dgrad kernels (run through gpurun from the repo root)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/convtprobe2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for op in convt gather; do
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR"; do
  tag=${op}_$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$tag -- python3 $REPO/tools/convt_micro.py --op $op --cin 64 --cout 32 --size 64 --reps 3 > $OUT/pmc_$tag.log 2>&1
done
done
