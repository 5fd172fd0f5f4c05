#!/bin/bash
# round 3, PNG encoder, first contact: parity of the HIP kernels with the host emulation, then sizes / times / per-kernel profile
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_png.py -m gpu -x -q -s > gpurun_out/r03png/pytest_png.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03png/pytest_png.log
tail -25 gpurun_out/r03png/pytest_png.log
timeout 300 python scripts/png_probe.py > gpurun_out/r03png/png_probe.json 2> gpurun_out/r03png/png_probe.err; tail -3 gpurun_out/r03png/png_probe.err
cat gpurun_out/r03png/png_probe.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03png/prof" -o png -- python "$GRAFT_REPO_ROOT/scripts/png_probe.py" 6 > /dev/null 2>&1)
find gpurun_out/r03png/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -20
