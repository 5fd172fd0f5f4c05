#!/bin/bash
# Quick GPU check: parity tests + both bench modes (+ optional CPU thread-scaling probe of the oracle).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^$") > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
python bench.py --steps 10 --warmup 2 --mode strict --cpu-seconds 0 > gpurun_out/bench_strict.json 2> gpurun_out/bench_strict.err
python bench.py --steps 20 --warmup 3 --mode fast --cpu-seconds 0 > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err
python - > gpurun_out/cpu_scaling.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, '.')
from oracle import c_oracle as co, scenes
ix = co.Index(None)
cfg = scenes.with_res(scenes.DEFAULT, 960, 540)
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
try: print(open('/sys/fs/cgroup/cpu.max').read())
except Exception as e: print(e)
for th in (1, 8, 32, 64, 128, 256):
    _, st = co.render(cfg, ix, threads=th)
    print(th, round(st['seconds'], 3), round(st['rays'] / st['seconds'] / 1e6, 3), 'Mray/s')
PY
grep -E "passed|failed|rsq|rc=" gpurun_out/pytest_gpu.log | tail; cat gpurun_out/cpu_scaling.txt
python - <<'PY'
import json
for m in ('strict','fast'):
    try:
        r=json.load(open(f'gpurun_out/bench_{m}.json')); print(m, round(r['value'],1),'Mpixel/s', round(r['kernel_ms'],3),'ms', 'frac',round(r['roofline']['frac'],3))
    except Exception as e: print(m, 'ERR', e, open(f'gpurun_out/bench_{m}.err').read()[-500:])
PY
