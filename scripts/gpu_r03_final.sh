#!/bin/bash
# round 3, evidence pass on the final tree: smoke, full GPU suite, every bench line (default with live PMC traffic, clustered sky, the
# d2h forms on their own, animation, every multi-GPU form smoke-run on this one-GPU box), rocprofv3 kernel stats of the default bench
# command on both skies and of the render -> bloom -> sRGB8 pipeline, PMC passes, parity report, per-config table, clustered fuzz.
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time timeout 900 python -m pytest tests -q -m gpu --durations=5 -rs) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --catalogue clustered --cpu-seconds 0 2> $O/bench_clustered.err | tail -n 1 > $O/bench_clustered.json
python bench.py --mode strict --cpu-seconds 0 --traffic static 2>/dev/null | tail -n 1 > $O/bench_strict.json
python bench.py --form batch --cpu-seconds 0 2>/dev/null | tail -n 1 > $O/bench_form_batch.json
python bench.py --form rgb8-batch --cpu-seconds 0 2>/dev/null | tail -n 1 > $O/bench_form_rgb8_batch.json
python bench.py --workload animation --steps 600 --cpu-seconds 0 2>/dev/null | tail -n 1 > $O/bench_c5_animation.json
for n in 2 4 8; do
  python bench.py --gpus $n --steps $((24 / n + 4)) --cpu-seconds 0 --sustained-frames 100 2>/dev/null | tail -n 1 > $O/bench_n${n}_single_process.json
done
for n in 2 8; do
  python bench.py --gpus $n --steps $((24 / n + 4)) --launcher torchrun --cpu-seconds 0 --sustained-frames 100 2>/dev/null | tail -n 1 > $O/bench_n${n}_torchrun_gloo.json
done
timeout 300 python scripts/configs_table.py 2> $O/configs_table.err | grep -v amdgpu > $O/configs_table.jsonl
timeout 600 python scripts/parity_report.py 2> $O/parity_report.err | grep -v amdgpu > $O/parity_report.jsonl
python scripts/fuzz_modes.py 20000 77 clustered 2>/dev/null > $O/fuzz_modes_clustered_20000.json
python scripts/fuzz_modes.py 20000 4242 2>/dev/null > $O/fuzz_modes_20000.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast -o fast -- python $R/bench.py --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 > $O/prof_fast.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_strict -o strict -- python $R/bench.py --mode strict --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 > $O/prof_strict.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_clustered -o clustered -- python $R/bench.py --catalogue clustered --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 > $O/prof_clustered.log 2>&1
for t in fast clustered; do
  S=synthetic; [ $t = clustered ] && S=clustered
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq_$t -o sq -- python $R/scripts/prof_frame.py --mode fast --stars $S --frames 3 > $O/pmc_sq_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2_$t -o sq2 -- python $R/scripts/prof_frame.py --mode fast --stars $S --frames 3 > $O/pmc_sq2_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_grbm_$t -o grbm -- python $R/scripts/prof_frame.py --mode fast --stars $S --frames 3 > $O/pmc_grbm_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$t -o fetch -- python $R/scripts/prof_frame.py --mode fast --stars $S --frames 3 > $O/pmc_fetch_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$t -o write -- python $R/scripts/prof_frame.py --mode fast --stars $S --frames 3 > $O/pmc_write_$t.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rgb8 -o rgb8 -- python $R/scripts/prof_rgb8.py > $O/prof_rgb8.log 2>&1
cd $R
tail -n 3 $O/smoke.log; tail -n 8 $O/pytest_gpu.log
for f in default clustered strict form_batch form_rgb8_batch c5_animation n2_single_process n4_single_process n8_single_process n2_torchrun_gloo n8_torchrun_gloo; do
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$f", round(d["value"], 1), round(d["ms_per_step"], 3), "frac", round(r["frac"], 3), "traffic", r.get("traffic"), str(r.get("traffic_kind"))[:20],
          "| d2h", {k: round(v["Mpixel_s"], 1) for k, v in d.get("with_d2h", {}).items()}, "| sustained", round(d.get("sustained", {}).get("ms_per_frame", 0), 3),
          "| rccl", d.get("rccl", {}).get("ranks"))
except Exception as e:
    print("$f NO JSON", e)
PY
done
cut -c1-400 $O/fuzz_modes_clustered_20000.json; cut -c1-400 $O/fuzz_modes_20000.json
head -3 $O/prof_fast/*kernel_stats.csv; head -3 $O/prof_clustered/*kernel_stats.csv; head -4 $O/prof_rgb8/*kernel_stats.csv
cat $O/parity_report.jsonl | cut -c1-300
