#!/bin/bash
# soak of the PNG paths (shared chip, partition forced, automatic) + the N > 1 smoke forms of bench.py with the png-batch leg
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03png
export TMPDIR=/tmp
timeout 200 python scripts/soak.py 50 11 > gpurun_out/r03png/soak_auto.txt 2>&1; tail -2 gpurun_out/r03png/soak_auto.txt
BLACKSTAR_POST_CUS=16 timeout 200 python scripts/soak.py 40 12 > gpurun_out/r03png/soak_16.txt 2>&1; tail -2 gpurun_out/r03png/soak_16.txt
BLACKSTAR_POST_CUS=8 timeout 200 python scripts/soak.py 30 13 > gpurun_out/r03png/soak_8.txt 2>&1; tail -2 gpurun_out/r03png/soak_8.txt
timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --sustained-frames 50 --cpu-seconds 0 > gpurun_out/r03png/bench_n2_single.json 2> gpurun_out/r03png/bench_n2_single.err; tail -2 gpurun_out/r03png/bench_n2_single.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --sustained-frames 50 --cpu-seconds 0 > gpurun_out/r03png/bench_n2_torchrun.json 2> gpurun_out/r03png/bench_n2_torchrun.err; tail -2 gpurun_out/r03png/bench_n2_torchrun.err
python - <<'PY'
import json
for f in ("bench_n2_single", "bench_n2_torchrun"):
    try:
        d = json.loads(open(f"gpurun_out/r03png/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], {k: round(v["Mpixel_s"], 1) for k, v in d.get("with_d2h", {}).items()}, d.get("rccl", {}).get("ranks"))
    except Exception as e:
        print(f, "FAILED", e)
PY
