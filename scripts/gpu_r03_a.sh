#!/bin/bash
# round 3, first pass: probe of page-locked memory kinds, full GPU suite, bench lines (default, clustered sky, every multi-GPU form
# smoke-run with N = 2 on this one-GPU box), rocprofv3 kernel stats of the default bench command.
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
python scripts/pinned_probe.py > $O/pinned_probe.txt 2>&1
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time timeout 900 python -m pytest tests -q -m gpu --durations=8) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --catalogue clustered --cpu-seconds 0 > $O/bench_clustered.json 2> $O/bench_clustered.err
python bench.py --gpus 2 --cpu-seconds 0 --sustained-frames 100 > $O/bench_n2_single_process.json 2> $O/bench_n2_single_process.err
python bench.py --gpus 2 --launcher torchrun --cpu-seconds 0 --sustained-frames 100 > $O/bench_n2_torchrun_gloo.json 2> $O/bench_n2_torchrun_gloo.err
python bench.py --gpus 4 --steps 10 --cpu-seconds 0 --sustained-frames 50 > $O/bench_n4_single_process.json 2> $O/bench_n4_single_process.err
python bench.py --gpus 8 --steps 6 --cpu-seconds 0 --sustained-frames 50 > $O/bench_n8_single_process.json 2> $O/bench_n8_single_process.err
python bench.py --gpus 8 --steps 6 --launcher torchrun --cpu-seconds 0 --sustained-frames 50 > $O/bench_n8_torchrun_gloo.json 2> $O/bench_n8_torchrun_gloo.err
python bench.py --form rgb8-batch --cpu-seconds 0 > $O/bench_form_rgb8_batch.json 2> $O/bench_form_rgb8_batch.err
python bench.py --form batch --cpu-seconds 0 > $O/bench_form_batch.json 2> $O/bench_form_batch.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast -o fast -- python $R/bench.py --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 > $O/prof_fast.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clustered -o clustered -- python $R/bench.py --catalogue clustered --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 > $O/prof_clustered.log 2>&1
cd $R
tail -n 3 $O/smoke.log; tail -n 25 $O/pytest_gpu.log
for f in default clustered n2_single_process n2_torchrun_gloo n4_single_process n8_single_process n8_torchrun_gloo form_rgb8_batch form_batch; do
  echo "== $f"; tail -n 3 $O/bench_$f.err | grep -v amdgpu.ids
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus")}, "frac", d["roofline"]["frac"])
    for k in ("with_d2h", "sustained", "rccl"):
        if k in d: print(" ", k, json.dumps(d[k])[:900])
    if "cpu_baseline" in d: print("  cpu", json.dumps(d["cpu_baseline"])[:600])
except Exception as e:
    print("NO JSON", e)
PY
done
head -5 $O/prof_fast/*kernel_stats.csv; head -5 $O/prof_clustered/*kernel_stats.csv
cat $O/pinned_probe.txt | cut -c1-400
