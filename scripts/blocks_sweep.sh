#!/bin/bash
# Resident workgroups per CU (BLACKSTAR_BLOCKS_PER_CU: 4 wavefronts each, i.e. waves per SIMD) against the sustained bench.
mkdir -p gpurun_out; rm -f gpurun_out/blocks.txt
for round in 1 2; do
  for b in ${BLOCKS:-4 3 2 1}; do
    echo -n "$round blocks_per_cu=$b " >> gpurun_out/blocks.txt
    BLACKSTAR_BLOCKS_PER_CU=$b timeout 200 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['kernel_ms'],4), round(r['value'],1))" >> gpurun_out/blocks.txt
  done
done
cat gpurun_out/blocks.txt
