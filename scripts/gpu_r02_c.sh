#!/bin/bash
# round 2 evidence run: full GPU suite, bench (both modes), rocprofv3 kernel stats + PMC passes of the trace kernel, kernel stats and
# HBM counters of the render -> bloom -> sRGB8 pipeline.  Raw output -> gpurun_out/ (scratch); scripts/collect_profiles.py r02 condenses.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(time timeout 900 python -m pytest tests -q -m gpu --durations=8) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
python scripts/ubench.py > gpurun_out/ubench.json 2> gpurun_out/ubench.err
python bench.py --steps 20 --warmup 3 --mode strict --cpu-seconds 0 --no-boundary > gpurun_out/bench_strict.json 2> gpurun_out/bench_strict.err
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp
for m in fast strict; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -o $m -- python $R/bench.py --steps 10 --warmup 2 --mode $m --cpu-seconds 0 --no-boundary > $R/gpurun_out/prof_$m.log 2>&1
done
for m in fast strict; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_$m -o sq -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_sq_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_sq2_$m -o sq2 -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_sq2_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/pmc_grbm_$m -o grbm -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_grbm_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$m -o fetch -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_fetch_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$m -o write -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_write_$m.log 2>&1
done
# the post pipeline: kernel stats, then HBM counters of the sweeps (separate passes)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rgb8 -o rgb8 -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/prof_rgb8.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_rgb8 -o fetch -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/pmc_fetch_rgb8.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_rgb8 -o write -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/pmc_write_rgb8.log 2>&1
cd $R
tail -n 14 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_strict.json | cut -c1-400; cat gpurun_out/bench_default.json | cut -c1-600; cat gpurun_out/prof_rgb8/rgb8_kernel_stats.csv
