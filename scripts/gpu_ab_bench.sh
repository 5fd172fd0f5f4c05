#!/bin/bash
# A/B of two library builds (same ABI) on the SUSTAINED bench (bench.py --steps 30), interleaved rounds; FAST and STRICT.
set -u
mkdir -p gpurun_out
rm -f gpurun_out/abb.txt
for round in 1 2 3; do
  for lib in ${LIBS:-blackstar_amd/libblackstar_gpu.so variants_prev.so}; do
    for m in ${MODES:-fast}; do
      echo -n "$round $lib $m " >> gpurun_out/abb.txt
      BLACKSTAR_LIB=$PWD/$lib timeout 200 python bench.py --steps 30 --warmup 5 --mode $m --cpu-seconds 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['kernel_ms'],4), round(r['value'],1))" >> gpurun_out/abb.txt
    done
  done
done
cat gpurun_out/abb.txt
