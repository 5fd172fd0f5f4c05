#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4) > gpurun_out/pytest_gpu.log 2>&1
for m in fast; do
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$m -o write -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_write_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$m -o fetch -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_fetch_$m.log 2>&1
done
python scripts/prof_frame.py --mode fast --frames 8 | tail -1
python scripts/prof_frame.py --mode strict --frames 8 | tail -1
python scripts/prof_frame.py --mode fast --frames 4 --scene default.yaml --stars none | tail -1
cat gpurun_out/pytest_gpu.log
python - <<'PY'
import csv
for g in ('write','fetch'):
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(f'gpurun_out/pmc_{g}_fast/{g}_counter_collection.csv')) if 'trace_frame' in r['Kernel_Name']]
    print(g, sum(v)/len(v), 'KB')
PY
