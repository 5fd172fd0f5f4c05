#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_png.py tests/test_gpu_parity.py -m gpu -x -q -k "png or animation or scene_directory" > gpurun_out/r03png/pytest_png3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03png/pytest_png3.log
tail -12 gpurun_out/r03png/pytest_png3.log
timeout 600 python scripts/png_animation_probe.py 240 > gpurun_out/r03png/png_animation.json 2> gpurun_out/r03png/png_animation.err; tail -3 gpurun_out/r03png/png_animation.err
cat gpurun_out/r03png/png_animation.json
