#!/bin/bash
# PMC passes on the bloom sweeps (through prof_rgb8.py = bs_render_rgb8 on the C3 frame).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bloom_stats -o b -- python scripts/prof_rgb8.py > gpurun_out/bloom_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/bloom_pmc1 -o b -- python scripts/prof_rgb8.py > gpurun_out/bloom_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR --output-format csv -d gpurun_out/bloom_pmc2 -o b -- python scripts/prof_rgb8.py > gpurun_out/bloom_pmc2.log 2>&1
python - <<'PY'
import csv, collections, glob
for d in ("bloom_pmc1", "bloom_pmc2"):
    fn = glob.glob(f"gpurun_out/{d}/*counter_collection.csv")
    if not fn: print(d, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn[0])):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "blur" in k or "bloom" in k:
            print(d, k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()})
fn = glob.glob("gpurun_out/bloom_stats/*kernel_stats.csv")
if fn: print(open(fn[0]).read()[:1500])
PY
