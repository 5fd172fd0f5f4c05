#!/bin/bash
# round 2, after the rotating-chain sweep: full GPU suite, bloom A/B, probe, kernel stats + HBM counters of the rgb8 pipeline, bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(time timeout 900 python -m pytest tests -q -m gpu --durations=5) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 100 python scripts/bloom_ab.py 2>&1 | grep -v amdgpu > gpurun_out/bloom_ab_final.txt
timeout 60 scripts/probe/sweep_probe > gpurun_out/sweep_probe_final.txt 2>&1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rgb8 -o rgb8 -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/prof_rgb8.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_rgb8 -o fetch -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/pmc_fetch_rgb8.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_rgb8 -o write -- python $R/scripts/prof_rgb8.py > $R/gpurun_out/pmc_write_rgb8.log 2>&1
cd $R
tail -n 10 gpurun_out/pytest_gpu.log; cat gpurun_out/bloom_ab_final.txt; cat gpurun_out/sweep_probe_final.txt; cat gpurun_out/prof_rgb8/rgb8_kernel_stats.csv; python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], json.dumps(d['boundary']))"
