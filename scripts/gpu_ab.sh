#!/bin/bash
# A/B of library builds (same ABI): kernel time of the C3 frame per variant, interleaved rounds.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python scripts/ubench.py > gpurun_out/ubench.json 2>&1
for round in 1 2 3; do
  for lib in blackstar_amd/libblackstar_gpu.so variants_w5.so variants_w6.so variants_w8.so; do
    for m in fast strict; do
      echo -n "$round $lib $m " >> gpurun_out/ab.txt
      BLACKSTAR_LIB=$PWD/$lib python scripts/prof_frame.py --mode $m --frames 6 | grep -o "'kernel_ms': [0-9.]*" >> gpurun_out/ab.txt
    done
  done
done
cat gpurun_out/ubench.json; cat gpurun_out/ab.txt
