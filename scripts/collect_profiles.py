#!/usr/bin/env python3
"""Condense gpurun_out/ (scratch) into profiles/<round>_* (tracked): rocprofv3 kernel stats + PMC summaries."""
import collections
import csv
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
summary = {}
for m in ("fast", "strict"):
    src = os.path.join(G, f"prof_{m}", f"{m}_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{R}_{m}_kernel_stats.csv"))
    pmc = {}
    for grp in ("sq", "sq2", "grbm", "fetch", "write"):
        fn = os.path.join(G, f"pmc_{grp}_{m}", f"{grp}_counter_collection.csv")
        if not os.path.exists(fn):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if "trace_frame" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc["VGPR_Count"], pmc["SGPR_Count"], pmc["LDS_Block_Size"] = r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")
        for k, v in agg.items():
            pmc[k] = sum(v) / len(v)
        kt = os.path.join(G, f"pmc_{grp}_{m}", f"{grp}_kernel_trace.csv")
        if os.path.exists(kt):
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "trace_frame" in r["Kernel_Name"]]
            if d:
                pmc[f"kernel_ns_in_{grp}_pass"] = sum(d) / len(d)
    summary[m] = pmc
for f in ("bench_default.json", "bench_strict.json", "ubench.json"):
    if os.path.exists(os.path.join(G, f)):
        shutil.copy(os.path.join(G, f), os.path.join(P, f"{R}_{f}"))
json.dump(summary, open(os.path.join(P, f"{R}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
