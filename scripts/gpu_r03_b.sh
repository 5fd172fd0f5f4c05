#!/bin/bash
# round 3, second pass: full GPU suite after the page-locked-range fix, the d2h forms on their own, torchrun smoke, PMC passes of the
# trace kernel on the uniform and on the clustered sky, kernel stats of the render -> bloom -> sRGB8 pipeline (bloom baseline).
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
(time timeout 900 python -m pytest tests -q -m gpu --durations=5 -rs) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python bench.py --form rgb8-batch --cpu-seconds 0 > $O/bench_form_rgb8_batch.json 2> $O/bench_form_rgb8_batch.err
python bench.py --form batch --cpu-seconds 0 > $O/bench_form_batch.json 2> $O/bench_form_batch.err
python bench.py --gpus 2 --launcher torchrun --cpu-seconds 0 --sustained-frames 100 > $O/bench_n2_torchrun_gloo.json 2> $O/bench_n2_torchrun_gloo.err
cd /tmp
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
tail -n 12 $O/pytest_gpu.log
for f in form_rgb8_batch form_batch n2_torchrun_gloo; do
  echo "== $f"
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus")}, "frac", d["roofline"]["frac"])
    for k in ("with_d2h", "sustained", "rccl"):
        if k in d: print(" ", k, json.dumps(d[k])[:700])
except Exception as e:
    print("NO JSON", e)
PY
done
cat $O/prof_rgb8/rgb8_kernel_stats.csv
python scripts/collect_profiles.py r03tmp r03 > /dev/null 2>&1; cat profiles/r03tmp_pmc_summary.json
