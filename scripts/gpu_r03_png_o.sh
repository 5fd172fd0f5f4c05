#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03png; mkdir -p $O
timeout 600 python -m pytest tests/test_png.py -m gpu -x -q > $O/pytest_png.log 2>&1; echo "pytest rc=$?" >> $O/pytest_png.log; tail -3 $O/pytest_png.log
timeout 300 python scripts/png_phase_probe.py > $O/png_phases.json 2> $O/png_phases.err; tail -2 $O/png_phases.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o png -- python "$GRAFT_REPO_ROOT/scripts/png_probe.py" 6 > /dev/null 2>&1)
grep png_ $O/prof/png_kernel_stats.csv
