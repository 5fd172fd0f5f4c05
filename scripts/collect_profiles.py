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
G, P = os.path.join(ROOT, "gpurun_out", *sys.argv[2:3]), os.path.join(ROOT, "profiles")  # optional 2nd argument: sub-directory of gpurun_out
os.makedirs(P, exist_ok=True)
summary = {}
for m in ("fast", "strict", "clustered"):  # "clustered": FAST mode on the non-uniform sky (bench.py --catalogue clustered)
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
    if pmc or m != "clustered":
        summary[m] = pmc
for f in ("bench_default.json", "bench_strict.json", "ubench.json"):
    if os.path.exists(os.path.join(G, f)):
        shutil.copy(os.path.join(G, f), os.path.join(P, f"{R}_{f}"))
json.dump(summary, open(os.path.join(P, f"{R}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

# ---- render -> bloom -> sRGB8 pipeline (scripts/prof_rgb8.py): kernel stats + FETCH_SIZE / WRITE_SIZE per kernel ----
src = os.path.join(G, "prof_rgb8", "rgb8_kernel_stats.csv")
if os.path.exists(src):
    shutil.copy(src, os.path.join(P, f"{R}_rgb8_kernel_stats.csv"))
SHORT = (("copyBuffer", "copyBuffer"), ("fillBuffer", "fillBuffer"), ("trace_frame_kernel", "trace_frame_kernel"), ("box_blur_sweep_rot", "box_blur_sweep_rot"),
         ("box_blur_sweep_lds", "box_blur_sweep_lds"), ("bloom_combine_srgb8", "bloom_combine_srgb8"), ("srgb8_kernel", "srgb8_kernel"))
rgb8 = collections.OrderedDict()
for grp, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    fn = os.path.join(G, f"pmc_{grp}_rgb8", f"{grp}_counter_collection.csv")
    if not os.path.exists(fn):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != counter:
            continue
        for key, name in SHORT:
            if key in r["Kernel_Name"]:
                agg[name].append(float(r["Counter_Value"]))
                break
    for name, v in agg.items():
        d = rgb8.setdefault(name, {})
        d[f"{counter}_KiB_per_launch"] = sum(v) / len(v)
        d[f"{counter}_launches"] = len(v)
for name, d in rgb8.items():
    if "FETCH_SIZE_KiB_per_launch" in d and "WRITE_SIZE_KiB_per_launch" in d:
        d["hbm_bytes_per_launch (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction)"] = 1024 * (2 * d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"])
if rgb8:
    rgb8["_note"] = ("scripts/prof_rgb8.py under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); a 1080p sweep reads and "
                     "writes 49.8 MB each = 99.5 MB algorithmic")
    json.dump(rgb8, open(os.path.join(P, f"{R}_rgb8_pmc_summary.json"), "w"), indent=1)
for src, dst in (("bloom_ab_final.txt", "bloom_ab.txt"), ("sweep_probe_final.txt", "sweep_probe.txt"), ("configs_table.jsonl", "configs_table.jsonl"),
                 ("bench_c5_animation.json", "bench_c5_animation.json")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{R}_{dst}"))

# ---- round 4: the bench lines of scripts/gpu_r04_final.sh, the partition trial A/B (one file per box), the fuzz re-run ----
import glob  # noqa: E402
for fn in sorted(glob.glob(os.path.join(G, "bench_*.json"))):
    dst = os.path.join(P, f"{R}_{os.path.basename(fn)}")
    if os.path.getsize(fn) > 2 and not os.path.exists(dst):
        shutil.copy(fn, dst)
for src, dst in (("fuzz_modes_20000.json", "fuzz_modes_20000.json"),):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{R}_{dst}"))
ab = os.path.join(G, "partition_trial_ab.jsonl")
if os.path.exists(ab):
    last = open(ab).read().strip().splitlines()[-1]
    try:
        host = json.loads(last).get("hostname", "box")
    except ValueError:
        host = "box"
    shutil.copy(ab, os.path.join(P, f"{R}_partition_trial_ab_{host[-8:]}.jsonl"))
