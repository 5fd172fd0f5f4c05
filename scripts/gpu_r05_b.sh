#!/bin/bash
# round 5, evidence pass (then: python scripts/collect_profiles.py r05 r05).  The GPU suite with the round's new tests, the driver-form bench
# line, every single-GPU BASELINE workload as the timed one, STRICT, the animation, the RCCL branch at world 1, N = 2 / 8 smoke lines on this
# one-GPU box (oversubscribed), rocprofv3 kernel stats of the default bench command in both modes, PMC passes for BOTH modes (each its own
# run), the pipeline's kernel stats.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time timeout 1500 python -m pytest tests -q -m gpu --durations=8 -rs) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python -m pytest tests -q -m gpu -s -k "worst_fuzz or ranks_like" 2>&1 | grep -E "^\{|FAST vs oracle|passed|failed" > $O/pytest_new_tests.txt
(time python bench.py) > $O/bench_default.json 2> $O/bench_default.err
python bench.py --workload default --cpu-seconds 0 --sustained-frames 100 2>/dev/null | tail -n 1 > $O/bench_c2_default.json
python bench.py --workload lensing-4k --cpu-seconds 0 --sustained-frames 100 2>/dev/null | tail -n 1 > $O/bench_c4_lensing_4k.json
python bench.py --mode strict --cpu-seconds 0 --traffic static --sustained-frames 100 2>/dev/null | tail -n 1 > $O/bench_strict.json
python bench.py --workload animation --steps 600 --cpu-seconds 0 2>/dev/null | tail -n 1 > $O/bench_c5_animation.json
python bench.py --gpus 1 --launcher torchrun --gather --cpu-seconds 0 --sustained-frames 100 --traffic static 2>$O/bench_rccl_world1.err | tail -n 1 > $O/bench_rccl_world1.json
python bench.py --gpus 1 --launcher torchrun --form split --cpu-seconds 0 2>/dev/null | tail -n 1 > $O/bench_rccl_world1_split.json
python bench.py --gpus 1 --launcher torchrun --workload animation --steps 48 --cpu-seconds 0 --sustained-frames 0 2>/dev/null | tail -n 1 > $O/bench_rccl_world1_animation.json
for n in 2 8; do
  s=$((24 / n + 4))
  python bench.py --gpus $n --steps $s --cpu-seconds 0 --sustained-frames 100 2>/dev/null | tail -n 1 > $O/bench_n${n}_single_process_smoke.json
  python bench.py --gpus $n --form split --cpu-seconds 0 2>/dev/null | tail -n 1 > $O/bench_n${n}_split_single_process_smoke.json
  python bench.py --gpus $n --steps $s --launcher torchrun --cpu-seconds 0 --sustained-frames 100 2>/dev/null | tail -n 1 > $O/bench_n${n}_torchrun_gloo_smoke.json
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast -o fast -- python $R/bench.py --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 --no-validate > $O/prof_fast.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_strict -o strict -- python $R/bench.py --mode strict --cpu-seconds 0 --no-boundary --form resident --sustained-frames 0 --no-validate > $O/prof_strict.log 2>&1
for t in fast strict; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq_$t -o sq -- python $R/scripts/prof_frame.py --mode $t --stars synthetic --frames 3 > $O/pmc_sq_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2_$t -o sq2 -- python $R/scripts/prof_frame.py --mode $t --stars synthetic --frames 3 > $O/pmc_sq2_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_grbm_$t -o grbm -- python $R/scripts/prof_frame.py --mode $t --stars synthetic --frames 3 > $O/pmc_grbm_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$t -o fetch -- python $R/scripts/prof_frame.py --mode $t --stars synthetic --frames 3 > $O/pmc_fetch_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$t -o write -- python $R/scripts/prof_frame.py --mode $t --stars synthetic --frames 3 > $O/pmc_write_$t.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rgb8 -o rgb8 -- python $R/scripts/prof_rgb8.py > $O/prof_rgb8.log 2>&1
cd $R
tail -n 3 $O/smoke.log; tail -n 12 $O/pytest_gpu.log; cat $O/pytest_new_tests.txt
for f in default c2_default c4_lensing_4k strict c5_animation rccl_world1 rccl_world1_split rccl_world1_animation n2_single_process_smoke n2_split_single_process_smoke n2_torchrun_gloo_smoke n8_single_process_smoke n8_split_single_process_smoke n8_torchrun_gloo_smoke; do
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/bench_$f.json").read().strip().splitlines() if l.startswith("{")][-1])
    r = d.get("roofline", {})
    sp = (d.get("with_d2h") or {}).get("split") or {}
    pr = sp.get("prediction_8_gpus") or {}
    print("$f", round(d["value"], 1), round(d["ms_per_step"], 3), "frac", round(r.get("frac", 0), 3), "valid", d.get("valid"), "rccl", (d.get("rccl") or {}).get("version"), "gather_ms", d.get("gather_ms"),
          "| d2h", {k: (round(v.get("Mpixel_s", 0), 1), v.get("frames_identical", v.get("identical_to_one_device"))) for k, v in d.get("with_d2h", {}).items() if isinstance(v, dict)},
          "| per_config", {k: (round(v["ms"], 3), round(v["frac"], 3)) for k, v in d.get("per_config", {}).items() if "ms" in v},
          "| split", sp.get("speedup_vs_one_device"), {k: pr.get(k) for k in ("steps_max_over_mean", "work_bound", "kernel_bound", "predicted_speedup_bound", "fixed_ms_per_band", "band_kernel_ms")})
    for p in (d.get("cpu_baseline") or {}).get("parity", []):
        print("   parity", p.get("mode"), p.get("config", "")[:50], p.get("outside_1e-4"), p.get("max_rel_where_ref>1e-3"), p.get("steps_equal"), p.get("fates_equal"))
except Exception as e:
    print("$f NO JSON", e)
PY
done
head -3 $O/prof_fast/*kernel_stats.csv; head -3 $O/prof_strict/*kernel_stats.csv; head -5 $O/prof_rgb8/*kernel_stats.csv
