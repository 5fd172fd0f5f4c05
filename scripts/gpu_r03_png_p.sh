#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03png; mkdir -p $O
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
timeout 300 python bench.py --form png-files --cpu-seconds 0 > $O/bench_form_png_files.json 2> $O/bench_form_png_files.err; tail -2 $O/bench_form_png_files.err
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --sustained-frames 50 --cpu-seconds 0 > $O/bench_n2_single.json 2> $O/bench_n2_single.err; tail -2 $O/bench_n2_single.err
timeout 300 python bench.py --gpus 2 --launcher torchrun --steps 6 --warmup 2 --cpu-seconds 0 --sustained-frames 50 > $O/bench_n2_torchrun_gloo.json 2> $O/bench_n2_torchrun_gloo.err; tail -2 $O/bench_n2_torchrun_gloo.err
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partitioned" > $O/pytest_part.log 2>&1; tail -2 $O/pytest_part.log
python - <<'PY'
import json
for f in ("bench_default", "bench_form_png_files", "bench_n2_single", "bench_n2_torchrun_gloo"):
    try:
        d = json.loads(open(f"gpurun_out/r03png/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), {k: (round(v["Mpixel_s"], 1), round(v["ms_per_frame_per_gpu"], 3)) for k, v in d.get("with_d2h", {}).items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
