#!/bin/bash
# Same-box interleaved A/B of the whole training step: the product library against a variant build (HDF_LIB_PATH).
# usage: tools/step_ab.sh <variant .so> [pairs]
cd "$(dirname "$0")/.."
VAR=$1
PAIRS=${2:-3}
for i in $(seq $PAIRS); do
  echo -n "product: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['hip_event_ms_per_step']['median'])"
  echo -n "variant: "; HDF_LIB_PATH=$VAR python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['hip_event_ms_per_step']['median'])"
done
