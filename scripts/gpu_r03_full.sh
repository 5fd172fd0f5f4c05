#!/bin/bash
# the whole GPU suite, then the PNG probes and the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03png/pytest_gpu_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03png/pytest_gpu_all.log
tail -6 gpurun_out/r03png/pytest_gpu_all.log
timeout 300 python scripts/png_phase_probe.py > gpurun_out/r03png/png_phases.json 2> gpurun_out/r03png/png_phases.err; tail -3 gpurun_out/r03png/png_phases.err
timeout 300 python scripts/png_probe.py > gpurun_out/r03png/png_probe.json 2> gpurun_out/r03png/png_probe.err; tail -3 gpurun_out/r03png/png_probe.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03png/prof" -o png -- python "$GRAFT_REPO_ROOT/scripts/png_probe.py" 6 > /dev/null 2>&1)
grep png_ gpurun_out/r03png/prof/png_kernel_stats.csv
timeout 600 python scripts/png_partition_ab.py > gpurun_out/r03png/png_partition_ab.jsonl 2> gpurun_out/r03png/png_partition_ab.err
cat gpurun_out/r03png/png_partition_ab.jsonl
