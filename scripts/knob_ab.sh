#!/bin/bash
# One binary, one environment knob of the library (BLACKSTAR_STAGGER, BLACKSTAR_LATE_POP_SLOT, BLACKSTAR_BLOCKS_PER_CU, ...), interleaved on the
# C3 / C2 / C4 frames: no rebuild, so nothing moves but the knob.   Usage: knob_ab.sh NAME VALUE [VALUE ...]   (ROUNDS=3, WORKLOADS=...)
set -u
cd "$(dirname "$0")/.."
name=$1; shift
for i in $(seq 1 "${ROUNDS:-3}"); do
  for v in "$@"; do
    for wl in ${WORKLOADS:-default-aa default lensing-4k}; do
      env "$name=$v" python bench.py --workload "$wl" --cpu-seconds 0 --traffic static --form resident --no-boundary --sustained-frames 0 --no-validate --steps 30 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$i $name=$v $wl', round(d['ms_per_step'], 4), round(d['value'], 1))"
    done
  done
done
