#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_png.py tests/test_gpu_parity.py -m gpu -x -q -k "png or animation or cpp_host or partition" > gpurun_out/r03png/pytest_png2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03png/pytest_png2.log
tail -5 gpurun_out/r03png/pytest_png2.log
timeout 600 python scripts/png_animation_probe.py 120 > gpurun_out/r03png/png_animation.json 2> gpurun_out/r03png/png_animation.err; tail -3 gpurun_out/r03png/png_animation.err
cat gpurun_out/r03png/png_animation.json
timeout 600 python bench.py > gpurun_out/r03png/bench_default.json 2> gpurun_out/r03png/bench_default.err; tail -3 gpurun_out/r03png/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03png/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, {k: (v.get("Mpixel_s"), v.get("ms_per_frame_per_gpu"), v.get("bytes_to_host_per_frame")) for k, v in d["with_d2h"].items()})
PY
