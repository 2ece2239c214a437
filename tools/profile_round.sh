#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02 [tag]
# Outputs under gpurun_out/prof_<round>/ ; tools/profile_summarise.py turns them into the files kept in profiles/.
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains).
set -u
ROUND=${1:-r04}
TAG=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${ROUND}${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=python3
# 1. kernel trace + stats of the bench command (3 timed steps): the build as it ships (three streams in a step), then
#    the SAME build with both stream knobs set (one stream: clean per-kernel durations), and the roofline leg alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- $PY $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_line_profiled.json 2> /dev/null
export HDF_NO_ASYNC_WGRAD=1 HDF_NO_BRANCH_OVERLAP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench1s -- $PY $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench1s_line_profiled.json 2> /dev/null
unset HDF_NO_ASYNC_WGRAD HDF_NO_BRANCH_OVERLAP
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roofline -- $PY $REPO/bench.py --roofline-only > $OUT/roofline_line.json 2> /dev/null
# 2. HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_conv_$c -- $PY $REPO/tools/conv_micro.py --cin 64 --cout 32 --xf 0 --reps 2 > /dev/null 2>&1
done
# 3. MFMA utilisation of the three conv kernel families (one pass: SQ + GRBM counters)
for shape in "conv 64 32 128" "conv 32 32 128" "wgrad 64 32 128" "wgrad 32 32 128" "conv 128 64 64" "conv 256 128 32"; do
  set -- $shape
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_$1_$2_$3_$4 -- $PY $REPO/tools/conv_micro.py --op $1 --cin $2 --cout $3 --size $4 --xf 0 --reps 5 > /dev/null 2>&1
done
# 4. per-kernel HBM traffic and matrix-pipe cycles of one whole step (one stream, so that a counter belongs to one kernel)
export HDF_NO_ASYNC_WGRAD=1 HDF_NO_BRANCH_OVERLAP=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_step_$c -- $PY $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_step_MFMA -- $PY $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
# 4b. LDS traffic per matrix instruction (VERDICT r03 #1a): LDS instructions, LDS-array cycles and bank-conflict cycles per kernel
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_step_LDS -- $PY $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
unset HDF_NO_ASYNC_WGRAD HDF_NO_BRANCH_OVERLAP
# 5. clock / power under the sustained roofline kernel
$PY $REPO/tools/clock_probe.py 64 32 > $OUT/clock_probe_64x32.txt 2>&1
$PY $REPO/tools/clock_probe.py 32 32 > $OUT/clock_probe_32x32.txt 2>&1
# 6. the un-profiled bench line
$PY $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_line_full.json 2> $OUT/bench_line_full.err
find $OUT -name "*.csv" | head -50
