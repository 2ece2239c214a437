#!/bin/bash
# Same-box interleaved A/B of the whole training step: the default arrangement against one with environment settings.
# usage: tools/env_ab.sh "HDF_NO_FUSED_APPLY=1" [pairs]
cd "$(dirname "$0")/.."
SET=$1
PAIRS=${2:-3}
line() { tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['hip_event_ms_per_step']['median'])"; }
for i in $(seq $PAIRS); do
  echo -n "default: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line
  echo -n "$SET: "; env $SET python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line
done
