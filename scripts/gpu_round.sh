#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench (both modes), rocprofv3 kernel trace + PMC passes.
set -u
R=${ROUND:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/nproc.txt; rocminfo | grep -E "Marketing|gfx9" | head -4 >> gpurun_out/nproc.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/nproc.txt 2>&1
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
(time timeout 400 python -m pytest tests -q -m gpu) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
python scripts/ubench.py > gpurun_out/ubench.json 2> gpurun_out/ubench.err
python bench.py --steps 20 --warmup 3 --mode strict --cpu-seconds 0 > gpurun_out/bench_strict.json 2> gpurun_out/bench_strict.err
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for m in fast strict; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -o $m -- python bench.py --steps 10 --warmup 2 --mode $m --cpu-seconds 0 > gpurun_out/prof_$m.log 2>&1
done
for m in fast strict; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq_$m -o sq -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_sq_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_sq2_$m -o sq2 -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_sq2_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_grbm_$m -o grbm -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_grbm_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$m -o fetch -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_fetch_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$m -o write -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_write_$m.log 2>&1
done
python scripts/prof_frame.py --mode fast --frames 8 > gpurun_out/wall_fast.txt 2>&1
tail -n 3 gpurun_out/smoke.log; tail -n 6 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_strict.json gpurun_out/bench_default.json; cat gpurun_out/wall_fast.txt
