#!/bin/bash
# round 2, evidence pass on the final kernels: smoke, full GPU suite, bench lines, rocprofv3 kernel stats + PMC passes of the trace
# kernel (both modes) and of the render -> bloom -> sRGB8 pipeline, bloom A/B, sweep probe, per-config table, animation workload.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
(time timeout 900 python -m pytest tests -q -m gpu --durations=5) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
python scripts/ubench.py > gpurun_out/ubench.json 2> gpurun_out/ubench.err
python bench.py --steps 20 --warmup 3 --mode strict --cpu-seconds 0 > gpurun_out/bench_strict.json 2> gpurun_out/bench_strict.err
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --workload animation --steps 600 --warmup 3 --cpu-seconds 0 > gpurun_out/bench_c5_animation.json 2> gpurun_out/bench_c5_animation.err
timeout 300 python scripts/configs_table.py 2> gpurun_out/configs_table.err | grep -v amdgpu > gpurun_out/configs_table.jsonl
timeout 100 python scripts/bloom_ab.py 2>&1 | grep -v amdgpu > gpurun_out/bloom_ab_final.txt
timeout 60 scripts/probe/sweep_probe > gpurun_out/sweep_probe_final.txt 2>&1
cd /tmp
for m in fast strict; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -o $m -- python $R/bench.py --mode $m --cpu-seconds 0 --no-boundary > $R/gpurun_out/prof_$m.log 2>&1
done
for m in fast strict; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_$m -o sq -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_sq_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_sq2_$m -o sq2 -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_sq2_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/pmc_grbm_$m -o grbm -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_grbm_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$m -o fetch -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_fetch_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$m -o write -- python $R/scripts/prof_frame.py --mode $m --frames 3 > $R/gpurun_out/pmc_write_$m.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rgb8 -o rgb8 -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/prof_rgb8.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_rgb8 -o fetch -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/pmc_fetch_rgb8.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_rgb8 -o write -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/pmc_write_rgb8.log 2>&1
cd $R
tail -n 2 gpurun_out/smoke.log; tail -n 10 gpurun_out/pytest_gpu.log; cat gpurun_out/configs_table.jsonl; cat gpurun_out/bloom_ab_final.txt | grep dma; grep -v "^ *wave" gpurun_out/sweep_probe_final.txt
cat gpurun_out/prof_rgb8/rgb8_kernel_stats.csv; cat gpurun_out/prof_fast/fast_kernel_stats.csv | head -5
python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']), json.dumps(d['boundary']), json.dumps(d['strict']))"
cat gpurun_out/bench_c5_animation.json
