#!/bin/bash
# Sweep of the first-tile phase offset (BLACKSTAR_STAGGER, shader cycles per SIMD slot): sustained bench per value.
set -u
mkdir -p gpurun_out
rm -f gpurun_out/stagger.txt
for round in 1 2; do
  for s in ${STAGGERS:-0 4000 8000 16000 32000 64000}; do
    echo -n "$round stagger=$s " >> gpurun_out/stagger.txt
    BLACKSTAR_STAGGER=$s timeout 200 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['kernel_ms'],4), round(r['value'],1))" >> gpurun_out/stagger.txt
  done
done
cat gpurun_out/stagger.txt
