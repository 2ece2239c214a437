#!/bin/bash
# Same-box interleaved comparison of the whole training step under several environment settings ("-" = default).
# usage: tools/env_abn.sh rounds "SET1" "SET2" ...     e.g. tools/env_abn.sh 5 - HDF_NO_FUSED_APPLY=1
cd "$(dirname "$0")/.."
R=$1; shift
line() { tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['hip_event_ms_per_step']['median'])"; }
for i in $(seq $R); do
  for SET in "$@"; do
    echo -n "$SET: "
    if [ "$SET" = "-" ]; then python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line
    else env $SET python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line; fi
  done
done
