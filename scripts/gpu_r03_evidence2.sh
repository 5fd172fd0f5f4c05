#!/bin/bash
# round 3, evidence pass on the final tree after the PNG work: smoke, the whole GPU suite, the PNG probes + kernel stats, the default
# bench line (live PMC traffic), the png-batch form on its own
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e2; mkdir -p $O
export TMPDIR=/tmp
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/png_phase_probe.py > $O/png_phases.json 2> $O/png_phases.err
timeout 300 python scripts/png_probe.py > $O/png_probe.json 2> $O/png_probe.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o png -- python "$GRAFT_REPO_ROOT/scripts/png_probe.py" 6 > /dev/null 2>&1)
timeout 600 python scripts/png_partition_ab.py > $O/png_partition_ab.jsonl 2> $O/png_partition_ab.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --form png-batch --cpu-seconds 0 > $O/bench_form_png_batch.json 2> $O/bench_form_png_batch.err
timeout 300 python bench.py --workload animation --steps 600 --cpu-seconds 0 > $O/bench_c5_animation.json 2> $O/bench_c5_animation.err
tail -n 3 $O/smoke.log; tail -n 4 $O/pytest_gpu.log; grep png_ $O/prof/png_kernel_stats.csv
python - <<'PY'
import json
for f in ("bench_default", "bench_form_png_batch", "bench_c5_animation"):
    try:
        d = json.loads(open(f"gpurun_out/r03e2/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), round(d["ms_per_step"], 3), {k: (round(v["Mpixel_s"], 1), v["bytes_to_host_per_frame"]) for k, v in d.get("with_d2h", {}).items() if "bytes_to_host_per_frame" in v}, d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 200 python scripts/soak.py 60 21 > $O/soak_auto.txt 2>&1; tail -1 $O/soak_auto.txt
timeout 600 python scripts/configs_table.py > $O/configs_table.jsonl 2> $O/configs_table.err; tail -2 $O/configs_table.err; cat $O/configs_table.jsonl | cut -c1-400
