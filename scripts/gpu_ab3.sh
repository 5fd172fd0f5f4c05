#!/bin/bash
# tests + stagger sweep of the persistent kernel vs the previous build
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6) > gpurun_out/pytest_gpu.log 2>&1
rm -f gpurun_out/ab3.txt
for round in 1 2; do
  for st in 0 8000 16000 26000 40000; do
    for m in fast strict; do
      echo -n "$round new stagger=$st $m " >> gpurun_out/ab3.txt
      BLACKSTAR_STAGGER=$st python scripts/prof_frame.py --mode $m --frames 8 | grep -o "'kernel_ms': [0-9.]*" >> gpurun_out/ab3.txt
    done
  done
  for m in fast strict; do
    echo -n "$round prev $m " >> gpurun_out/ab3.txt
    BLACKSTAR_LIB=$PWD/variants_prev.so python scripts/prof_frame.py --mode $m --frames 8 | grep -o "'kernel_ms': [0-9.]*" >> gpurun_out/ab3.txt
  done
done
for st in 0 26000; do
  echo -n "bench fast stagger=$st " >> gpurun_out/ab3.txt
  BLACKSTAR_STAGGER=$st python bench.py --steps 30 --warmup 5 --mode fast --cpu-seconds 0 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print(round(r['kernel_ms'],4), round(r['value'],1))" >> gpurun_out/ab3.txt
done
cat gpurun_out/pytest_gpu.log gpurun_out/ab3.txt
