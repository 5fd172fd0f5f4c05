#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r03png/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03png/smoke.log; tail -4 gpurun_out/r03png/smoke.log
timeout 300 python -m pytest tests/test_png.py -m gpu -x -q > gpurun_out/r03png/pytest_png.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03png/pytest_png.log; tail -3 gpurun_out/r03png/pytest_png.log
timeout 400 python scripts/png_fuzz.py 4000 > gpurun_out/r03png/png_fuzz.json 2> gpurun_out/r03png/png_fuzz.err; echo "fuzz rc=$?"; tail -2 gpurun_out/r03png/png_fuzz.err; cat gpurun_out/r03png/png_fuzz.json
