#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
timeout 300 python bench.py --gpus 2 --launcher torchrun --steps 6 --warmup 2 --cpu-seconds 0 --sustained-frames 50 > gpurun_out/r03png/bench_n2_torchrun_gloo.json 2> gpurun_out/r03png/bench_n2_torchrun_gloo.err; tail -2 gpurun_out/r03png/bench_n2_torchrun_gloo.err
timeout 300 python bench.py --gpus 8 --steps 6 --warmup 2 --cpu-seconds 0 --sustained-frames 50 > gpurun_out/r03png/bench_n8_single.json 2> gpurun_out/r03png/bench_n8_single.err; tail -2 gpurun_out/r03png/bench_n8_single.err
timeout 300 python bench.py --form png-batch --cpu-seconds 0 > gpurun_out/r03png/bench_form_png_batch.json 2> gpurun_out/r03png/bench_form_png_batch.err; tail -2 gpurun_out/r03png/bench_form_png_batch.err
python - <<'PY'
import json
for f in ("bench_n2_torchrun_gloo", "bench_n8_single", "bench_form_png_batch"):
    try:
        d = json.loads(open(f"gpurun_out/r03png/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), d["config"].get("image", "")[:60], {k: round(v["Mpixel_s"], 1) for k, v in d.get("with_d2h", {}).items()}, d.get("rccl", {}))
    except Exception as e:
        print(f, "FAILED", e)
PY
