mkdir -p gpurun_out; rm -f gpurun_out/w5.txt
for round in 1 2 3; do
  echo -n "$round w4 " >> gpurun_out/w5.txt
  timeout 200 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['kernel_ms'],4), round(r['value'],1))" >> gpurun_out/w5.txt
  echo -n "$round w5 " >> gpurun_out/w5.txt
  BLACKSTAR_LIB=$PWD/variants_w5.so BLACKSTAR_BLOCKS_PER_CU=5 timeout 200 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['kernel_ms'],4), round(r['value'],1))" >> gpurun_out/w5.txt
done
cat gpurun_out/w5.txt
