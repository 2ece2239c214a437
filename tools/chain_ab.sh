#!/bin/bash
# Same-box interleaved A/B of the training step: persistent transformer kernels (default) against the launch chain
# (HDF_NO_TF_CHAIN=1), plus a one-stream kernel trace of each for the per-kernel durations.
# usage: tools/chain_ab.sh [pairs]
cd "$(dirname "$0")/.."
PAIRS=${1:-3}
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['hip_event_ms_per_step']['median'])"; }
for i in $(seq $PAIRS); do
  echo -n "chain:    "; one
  echo -n "launches: "; HDF_NO_TF_CHAIN=1 one
done
