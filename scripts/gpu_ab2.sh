#!/bin/bash
# Sustained-load A/B (bench.py, back-to-back launches) of library variants + bloom test/timing.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests -q -m gpu -x -k "bloom or cpp_host" 2>&1 | tail -5) > gpurun_out/pytest_bloom.log 2>&1
rm -f gpurun_out/ab2.txt
for round in 1 2 3; do
  for lib in ${LIBS:-blackstar_amd/libblackstar_gpu.so variants_newton2.so}; do
    echo -n "$round $lib " >> gpurun_out/ab2.txt
    BLACKSTAR_LIB=$PWD/$lib python bench.py --steps 40 --warmup 5 --mode fast --cpu-seconds 0 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print(round(r['kernel_ms'],4), round(r['value'],1))" >> gpurun_out/ab2.txt
  done
done
bash scripts/gpu_extra.sh 2>&1 | tail -3
cat gpurun_out/pytest_bloom.log gpurun_out/ab2.txt
