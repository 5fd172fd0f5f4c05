#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03png; mkdir -p $O
timeout 600 python bench.py --cpu-seconds 0 --traffic static > $O/bench_q.json 2> $O/bench_q.err; tail -2 $O/bench_q.err
timeout 300 python bench.py --gpus 2 --launcher torchrun --steps 6 --warmup 2 --cpu-seconds 0 --sustained-frames 50 > $O/bench_q2.json 2> $O/bench_q2.err; tail -2 $O/bench_q2.err
python - <<'PY'
import json
for f in ("bench_q", "bench_q2"):
    d = json.loads(open(f"gpurun_out/r03png/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), d["with_d2h"]["png_files"])
PY
