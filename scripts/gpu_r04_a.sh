#!/bin/bash
# round 4, first GPU pass: smoke, the GPU suite (incl. the 2^32 counter test), the default bench line with per_config + validation, the new
# workloads, every form at N = 2 on this one-GPU box (smoke: oversubscribed), the FAST-vs-STRICT fuzz (the trace kernel file changed).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time timeout 1500 python -m pytest tests -q -m gpu --durations=8 -rs) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --workload default --cpu-seconds 0 --sustained-frames 100 2> $O/bench_c2.err | tail -n 1 > $O/bench_c2_default.json
python bench.py --workload lensing-4k --cpu-seconds 0 --sustained-frames 100 2> $O/bench_c4.err | tail -n 1 > $O/bench_c4_lensing_4k.json
python bench.py --form split --cpu-seconds 0 2> $O/bench_split1.err | tail -n 1 > $O/bench_n1_split.json
python bench.py --gpus 2 --steps 10 --cpu-seconds 0 --sustained-frames 100 2> $O/bench_n2.err | tail -n 1 > $O/bench_n2_single_process.json
python bench.py --gpus 2 --form split --cpu-seconds 0 2> $O/bench_n2_split.err | tail -n 1 > $O/bench_n2_split_single_process.json
python bench.py --gpus 2 --steps 10 --launcher torchrun --cpu-seconds 0 --sustained-frames 100 2> $O/bench_n2_tr.err | tail -n 1 > $O/bench_n2_torchrun_gloo.json
python bench.py --gpus 2 --form split --launcher torchrun --cpu-seconds 0 2> $O/bench_n2_split_tr.err | tail -n 1 > $O/bench_n2_split_torchrun_gloo.json
(time python scripts/fuzz_modes.py 20000 4242) 2> $O/fuzz.time > $O/fuzz_modes_20000.json
tail -n 3 $O/smoke.log; tail -n 14 $O/pytest_gpu.log
for f in default c2_default c4_lensing_4k n1_split n2_single_process n2_split_single_process n2_torchrun_gloo n2_split_torchrun_gloo; do
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("$f", round(d["value"], 1), round(d["ms_per_step"], 3), "frac", round(r.get("frac", 0), 3), "valid", d.get("valid"),
          "| d2h", {k: (round(v.get("Mpixel_s", 0), 1), v.get("frames_identical", v.get("identical_to_one_device"))) for k, v in d.get("with_d2h", {}).items() if isinstance(v, dict)},
          "| per_config", {k: (round(v["ms"], 3), round(v["frac"], 3)) for k, v in d.get("per_config", {}).items() if "ms" in v},
          "| split", d.get("split"))
    v = d.get("validation", {})
    print("   validation", {k: v.get(k) for k in ("frames_identical_across_devices", "steps_per_device", "repeat_identical_on_device0", "valid", "error")})
except Exception as e:
    print("$f NO JSON", e)
PY
done
cut -c1-500 $O/fuzz_modes_20000.json; tail -n 4 $O/fuzz.time
tail -n 5 $O/*.err | cut -c1-300
