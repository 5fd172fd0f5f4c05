#!/bin/bash
# round 5, first pass: smoke, the GPU suite, the driver-form bench line (now with cpu_baseline.parity and with_d2h.split.prediction_8_gpus),
# the RCCL branch of bench.py executed at world 1 (one rank under torch.distributed.run, --gather too), the 100 000-scene FAST-vs-STRICT
# fuzz on both skies with the worst scene NAMED (seeds of r04: 927 uniform, 2026 clustered).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time timeout 1500 python -m pytest tests -q -m gpu --durations=8 -rs) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
(time python bench.py) > $O/bench_default.json 2> $O/bench_default.err
(time python bench.py --gpus 1 --launcher torchrun --gather --cpu-seconds 0 --sustained-frames 100 --traffic static) > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
(time python bench.py --gpus 1 --launcher torchrun --form split --cpu-seconds 0) > $O/bench_rccl_world1_split.json 2> $O/bench_rccl_world1_split.err
(time python scripts/fuzz_modes.py 100000 927) 2> $O/fuzz_u.time > $O/fuzz_modes_100000.json
(time python scripts/fuzz_modes.py 100000 2026 clustered) 2> $O/fuzz_c.time > $O/fuzz_modes_clustered_100000.json
tail -n 3 $O/smoke.log; tail -n 12 $O/pytest_gpu.log
for f in default rccl_world1 rccl_world1_split; do
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/bench_$f.json").read().strip().splitlines() if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("$f", round(d["value"], 1), round(d["ms_per_step"], 3), "frac", round(r.get("frac", 0), 3), "valid", d.get("valid"), "rccl", d.get("rccl"), "gather_ms", d.get("gather_ms"))
    cb = d.get("cpu_baseline") or {}
    for p in cb.get("parity", []):
        print("  parity", p.get("mode"), p.get("config", "")[:60], {k: p.get(k) for k in ("values", "outside_1e-4", "max_abs", "max_rel_where_ref>1e-3", "bit_identical", "steps_equal", "fates_equal", "error")})
    sp = (d.get("with_d2h") or {}).get("split") or {}
    print("  split", sp.get("speedup_vs_one_device"), json.dumps(sp.get("prediction_8_gpus"))[:1200])
except Exception as e:
    print("$f NO JSON", e)
PY
  tail -n 5 $O/bench_$f.err
done
cut -c1-300 $O/fuzz_modes_100000.json; echo; python -c "import json;d=json.load(open('$O/fuzz_modes_100000.json'));print(d['worst_rel'], json.dumps(d['worst_rel_scene']))"
cut -c1-300 $O/fuzz_modes_clustered_100000.json; echo; python -c "import json;d=json.load(open('$O/fuzz_modes_clustered_100000.json'));print(d['worst_rel'], json.dumps(d['worst_rel_scene']))"
cat $O/fuzz_u.time $O/fuzz_c.time | grep real
