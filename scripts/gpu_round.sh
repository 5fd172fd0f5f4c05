#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench (both modes), rocprofv3 kernel trace + PMC passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/nproc.txt; rocminfo | grep -E "Marketing|gfx9" | head -4 >> gpurun_out/nproc.txt
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
(time timeout 1500 python -m pytest tests -q -m gpu) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
python bench.py --steps 10 --warmup 2 --mode strict --cpu-seconds 0 > gpurun_out/bench_strict.json 2> gpurun_out/bench_strict.err
python bench.py --steps 20 --warmup 3 --mode fast > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err
for m in fast strict; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -o $m -- python bench.py --steps 10 --warmup 2 --mode $m --cpu-seconds 0 > gpurun_out/prof_$m.log 2>&1
done
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
for m in fast strict; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq_$m -o sq -- python scripts/prof_frame.py --mode $m --frames 2 > gpurun_out/pmc_sq_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_grbm_$m -o grbm -- python scripts/prof_frame.py --mode $m --frames 2 > gpurun_out/pmc_grbm_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$m -o fetch -- python scripts/prof_frame.py --mode $m --frames 2 > gpurun_out/pmc_fetch_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$m -o write -- python scripts/prof_frame.py --mode $m --frames 2 > gpurun_out/pmc_write_$m.log 2>&1
done
find gpurun_out -name "*.csv" | head -40
tail -n 5 gpurun_out/smoke.log; tail -n 8 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_strict.json gpurun_out/bench_fast.json
