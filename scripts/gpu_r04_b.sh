#!/bin/bash
# round 4, second GPU pass (after the host-layer split, the debug library and the measured partition): smoke, GPU suite, the default bench
# line, N = 2 smoke lines, the FAST-vs-STRICT fuzz on the 64-bit-counter kernel, the partition trial A/B on the 26 combinations.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04b}
mkdir -p $O
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time timeout 1500 python -m pytest tests -q -m gpu --durations=8 -rs) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 2 --steps 20 --cpu-seconds 0 --sustained-frames 100 2> $O/bench_n2.err | tail -n 1 > $O/bench_n2_single_process.json
[ "${2:-fuzz}" = nofuzz ] || (time python scripts/fuzz_modes.py 20000 4242) 2> $O/fuzz.time > $O/fuzz_modes_20000.json
(time timeout 1200 python scripts/partition_trial_ab.py 36) 2> $O/partition_ab.err > $O/partition_trial_ab.jsonl
rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2 > $O/sclk.txt
tail -n 3 $O/smoke.log; tail -n 25 $O/pytest_gpu.log
for f in default n2_single_process; do
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("$f", round(d["value"], 1), round(d["ms_per_step"], 3), "frac", round(r.get("frac", 0), 3), "valid", d.get("valid"),
          "| d2h", {k: (round(v.get("Mpixel_s", 0), 1), v.get("frames_identical", v.get("identical_to_one_device"))) for k, v in d.get("with_d2h", {}).items() if isinstance(v, dict)},
          "| per_config", {k: (round(v["ms"], 3), round(v["frac"], 3)) for k, v in d.get("per_config", {}).items() if "ms" in v})
    v = d.get("validation", {})
    print("   validation", {k: v.get(k) for k in ("frames_identical_across_devices", "steps_per_device", "repeat_identical_on_device0", "valid", "error")})
    print("   sustained sclk", [(x.get("sclk_MHz_mean"), x.get("power_W_mean")) for x in (d.get("sustained", {}).get("device") or [])])
except Exception as e:
    print("$f NO JSON", e)
PY
done
cut -c1-400 $O/fuzz_modes_20000.json; tail -n 4 $O/fuzz.time
cat $O/partition_trial_ab.jsonl | cut -c1-420; tail -n 5 $O/partition_ab.err
