#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_png.py -m gpu -x -q > gpurun_out/r03png/pytest_png.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03png/pytest_png.log
tail -3 gpurun_out/r03png/pytest_png.log
timeout 600 python scripts/png_partition_ab.py > gpurun_out/r03png/png_partition_ab.jsonl 2> gpurun_out/r03png/png_partition_ab.err; tail -3 gpurun_out/r03png/png_partition_ab.err
cat gpurun_out/r03png/png_partition_ab.jsonl
