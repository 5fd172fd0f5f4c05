#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/ab4.txt
for round in 1 2 3; do
  echo -n "$round base4 " >> gpurun_out/ab4.txt
  python scripts/prof_frame.py --mode fast --frames 8 | grep -o "'kernel_ms': [0-9.]*" >> gpurun_out/ab4.txt
  for st in 0 13000; do
    echo -n "$round w5 stagger=$st " >> gpurun_out/ab4.txt
    BLACKSTAR_STAGGER=$st BLACKSTAR_BLOCKS_PER_CU=5 BLACKSTAR_LIB=$PWD/variants_w5.so python scripts/prof_frame.py --mode fast --frames 8 | grep -o "'kernel_ms': [0-9.]*" >> gpurun_out/ab4.txt
  done
done
cat gpurun_out/ab4.txt
