#!/bin/bash
# bench.py's CPU baseline (the oracle's fp32 train step on the host cores) at several thread counts: the stages the child
# finishes within its budget (4x64^3 crop, then 4x128^3 at batch 1 and 2); one JSON line per count.
# usage: tools/cpu_thread_sweep.sh [counts...]    (default 8 16 32 64)
cd "$(dirname "$0")/.."
for t in ${@:-8 16 32 64}; do
  echo -n "threads $t: "
  HIP_VISIBLE_DEVICES="" OMP_NUM_THREADS=$t HDF_BENCH_CPU_THREADS=$t timeout 300 python bench.py --cpu-baseline-child 2>/dev/null | tail -1 |
    python -c "import json,sys; l=sys.stdin.read().strip(); r=json.loads(l) if l else {}; print(r.get('value'), '|', r.get('sample'))"
done
