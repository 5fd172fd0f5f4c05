#!/bin/bash
# A/B of library builds (same ABI): kernel time of the C3 frame per variant, interleaved rounds, plus GPU tests.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout ${TEST_TIMEOUT:-300} python -m pytest tests -q -m gpu -x 2>&1 | tail -15) > gpurun_out/pytest_gpu.log 2>&1
rm -f gpurun_out/ab.txt
for round in 1 2 3; do
  for lib in ${LIBS:-blackstar_amd/libblackstar_gpu.so variants_prev.so}; do
    for m in fast strict; do
      echo -n "$round $lib $m " >> gpurun_out/ab.txt
      BLACKSTAR_LIB=$PWD/$lib timeout 120 python scripts/prof_frame.py --mode $m --frames 8 | grep -o "'kernel_ms': [0-9.]*" >> gpurun_out/ab.txt
    done
  done
done
timeout 300 python bench.py --steps 20 --warmup 3 --mode strict --cpu-seconds 0 > gpurun_out/bench_strict.json 2> gpurun_out/bench_strict.err
timeout 300 python bench.py --steps 20 --warmup 3 --mode fast --cpu-seconds 0 > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err
cat gpurun_out/pytest_gpu.log; cat gpurun_out/ab.txt
python - <<'PY'
import json
for m in ('strict','fast'):
    try:
        r=json.load(open(f'gpurun_out/bench_{m}.json')); print(m, round(r['value'],1),'Mpixel/s', round(r['kernel_ms'],3),'ms', 'frac',round(r['roofline']['frac'],3), 'lane_eff', round(r['lane_efficiency'],4))
    except Exception as e: print(m, 'ERR', e, open(f'gpurun_out/bench_{m}.err').read()[-800:])
PY
