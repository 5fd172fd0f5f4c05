#!/bin/bash
# A/B of trace-kernel builds on one GPU box: variants_w_<name>.so (built HERE with `make -C blackstar_amd/csrc EXTRA=... OUT=../../variants_w_<name>.so`,
# hipcc cross-compiles) timed interleaved on the C3 / C2 / C4 frames, three rounds.  Kernel-only, image resident, no CPU leg.
#   gpurun -- 'bash scripts/variant_ab.sh base fma3 > gpurun_out/ab.txt'
set -u
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-3}
WORKLOADS=${WORKLOADS:-default-aa default lensing-4k}
for i in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    for wl in $WORKLOADS; do
      BLACKSTAR_LIB=$PWD/variants_w_$v.so python bench.py --workload "$wl" --cpu-seconds 0 --traffic static --form resident --no-boundary \
        --sustained-frames 0 --no-validate --steps 30 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$i $v $wl', round(d['ms_per_step'], 4), round(d['value'], 1))"
    done
  done
done
