#!/bin/bash
# PNG encoder: where a block's time goes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
timeout 300 python scripts/png_phase_probe.py > gpurun_out/r03png/png_phases.json 2> gpurun_out/r03png/png_phases.err; tail -3 gpurun_out/r03png/png_phases.err
cat gpurun_out/r03png/png_phases.json
