#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03png/pytest_gpu_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03png/pytest_gpu_all.log
tail -5 gpurun_out/r03png/pytest_gpu_all.log
