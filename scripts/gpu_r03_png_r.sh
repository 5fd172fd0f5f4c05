#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03png; mkdir -p $O
timeout 300 python bench.py --gpus 4 --steps 6 --warmup 2 --cpu-seconds 0 --sustained-frames 50 > $O/bench_n4_single.json 2> $O/bench_n4_single.err
timeout 300 python bench.py --gpus 8 --steps 6 --warmup 2 --cpu-seconds 0 --sustained-frames 50 > $O/bench_n8_single.json 2> $O/bench_n8_single.err
timeout 400 python bench.py --gpus 8 --launcher torchrun --steps 6 --warmup 2 --cpu-seconds 0 --sustained-frames 50 > $O/bench_n8_torchrun_gloo.json 2> $O/bench_n8_torchrun_gloo.err
python - <<'PY'
import json
for f in ("bench_n4_single", "bench_n8_single", "bench_n8_torchrun_gloo"):
    try:
        d = json.loads(open(f"gpurun_out/r03png/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), {k: round(v.get("Mpixel_s", -1), 1) for k, v in d.get("with_d2h", {}).items()}, d.get("rccl", {}).get("ranks"))
    except Exception as e:
        print(f, "FAILED", e)
PY
