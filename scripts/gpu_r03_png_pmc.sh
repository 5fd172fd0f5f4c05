#!/bin/bash
# PMC counters of the PNG kernels (own run: counters only with --kernel-trace)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03pmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$GRAFT_REPO_ROOT/$O/p1" -o png -- python "$GRAFT_REPO_ROOT/scripts/png_probe.py" 4 > "$GRAFT_REPO_ROOT/$O/p1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_SMEM --output-format csv -d "$GRAFT_REPO_ROOT/$O/p2" -o png -- python "$GRAFT_REPO_ROOT/scripts/png_probe.py" 4 > "$GRAFT_REPO_ROOT/$O/p2.log" 2>&1
cd "$GRAFT_REPO_ROOT"
ls $O/p1 $O/p2 | head; tail -2 $O/p1.log
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r03pmc/p1", "gpurun_out/r03pmc/p2"):
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "png_" in k:
                acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "dispatches", len(next(iter(c.values()))))
PY
