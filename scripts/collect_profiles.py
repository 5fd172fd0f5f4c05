#!/usr/bin/env python3
"""Condense gpurun_out/<round>/ (scratch, written by scripts/gpu_round.sh on the GPU box) into the tracked profiles/<round>_*.

What makes profiles/ reproduce the driver's number: the rocprofv3 pass profiles the SAME command the driver runs, and
from its kernel trace the launches of the TIMED region are taken -- the FAST / STRICT trace kernel's dispatches number warmup .. warmup +
steps - 1 of the process, i.e. without the cold ones of the clock ramp -- and their median / min / mean set beside the `kernel_ms` the very
same process printed, with the shader clock and package power that process sampled (its `sustained.device`).  The rocprofv3 --stats table
(all launches of the process, every leg) is kept as it came, below '#' lines that say so.
Usage: collect_profiles.py r06"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def find(pattern):
    hits = sorted(glob.glob(os.path.join(G, pattern), recursive=True))
    return hits[0] if hits else None


def load(name):
    try:
        return json.load(open(os.path.join(G, name)))
    except (OSError, ValueError):
        return None


summary = {}
for mode in ("fast", "strict"):
    line = load(f"bench_profiled_{mode}.json")
    trace = find(f"prof_{mode}/**/{mode}_kernel_trace.csv")
    stats = find(f"prof_{mode}/**/{mode}_kernel_stats.csv")
    if not (line and trace and stats):
        continue
    want = "trace_frame_kernel<true>" if mode == "fast" else "trace_frame_kernel<false>"
    rows = [r for r in csv.DictReader(open(trace)) if want in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ns = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    w, k = int(line["warmup"]), int(line["steps"])
    timed, cold = ns[w:w + k], ns[:w]
    dev = ((line.get("sustained") or {}).get("device") or [None])[0] or {}
    s = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --traffic static --cpu-seconds 0" + (" --mode strict --form resident" if mode == "strict" else ""),
         "kernel": want, "launches_in_process": len(ns), "warmup_launches_dropped": w, "timed_launches": len(timed),
         "timed_ms": {"median": statistics.median(timed) / 1e6, "min": min(timed) / 1e6, "mean": statistics.fmean(timed) / 1e6, "max": max(timed) / 1e6},
         "warmup_ms": {"first": cold[0] / 1e6, "mean": statistics.fmean(cold) / 1e6} if cold else None,
         "bench_line_same_process": {"kernel_ms": line["kernel_ms"], "kernel_ms_median_of_each": statistics.median(line.get("kernel_ms_each") or [line["kernel_ms"]]),
                                     "kernel_ms_last_hipevent": line.get("kernel_ms_last_hipevent"), "ms_per_step": line["ms_per_step"], "value": line["value"],
                                     "roofline_frac": line["roofline"]["frac"]},
         "rocprof_median_over_bench_kernel_ms": statistics.median(timed) / 1e6 / line["kernel_ms"],
         "note": "kernel_ms = mean of HIP-event pairs around each launch (incl. the 64-byte counter memset and read-back nodes, ~10-35 us); "
                 "kernel_ms_last_hipevent = bs_stats' event pair around the kernel alone, last launch",
         "sclk_MHz_mean": dev.get("sclk_MHz_mean"), "power_W_mean": dev.get("power_W_mean"), "clock_source": "bench.py's DeviceSampler during the `sustained` leg of the same process"}
    summary[mode] = s
    with open(os.path.join(P, f"{R}_{mode}_kernel_stats.csv"), "w") as f:
        f.write(f"# {s['command']}\n")
        f.write(f"# timed launches of the process ({want}, dispatches {w}..{w + k - 1}; the {w} warm-up launches before them dropped): "
                f"median {s['timed_ms']['median']:.4f} ms, min {s['timed_ms']['min']:.4f}, mean {s['timed_ms']['mean']:.4f}, max {s['timed_ms']['max']:.4f}\n")
        f.write(f"# the same process printed kernel_ms {line['kernel_ms']:.4f} (ms_per_step {line['ms_per_step']:.4f}, {line['value']:.1f} Mpixel/s, frac {line['roofline']['frac']:.4f}); "
                f"sclk {dev.get('sclk_MHz_mean')} MHz, {dev.get('power_W_mean')} W (sampled during its sustained leg)\n")
        f.write("# below: rocprofv3 --stats over ALL launches of that process (warm-up, per_config, delivered forms, sustained, boundary legs)\n")
        f.write(open(stats).read())
    shutil.copy(os.path.join(G, f"bench_profiled_{mode}.json"), os.path.join(P, f"{R}_bench_profiled_{mode}.json"))

# PMC passes over scripts/prof_frame.py (3 launches per pass)
pmc_all = {}
for mode in ("fast", "strict"):
    pmc = {}
    for grp in ("sq", "sq2", "fetch", "write"):
        fn = find(f"pmc_{grp}_{mode}/**/{grp}_counter_collection.csv")
        if not fn:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if "trace_frame" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc["VGPR_Count"], pmc["SGPR_Count"], pmc["LDS_Block_Size"] = r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")
        for k_, v in agg.items():
            pmc[k_] = sum(v) / len(v)
        kt = find(f"pmc_{grp}_{mode}/**/{grp}_kernel_trace.csv")
        if kt:
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "trace_frame" in r["Kernel_Name"]]
            if d:
                pmc[f"kernel_ns_in_{grp}_pass"] = sum(d) / len(d)
    if pmc:
        if "GRBM_GUI_ACTIVE" in pmc and summary.get(mode):
            line = load(f"bench_profiled_{mode}.json")
            flop = line["roofline"]["flop_per_launch"]
            cycles = pmc["GRBM_GUI_ACTIVE"] / 8
            pmc["frac_cycles"] = flop / (cycles * 256 * 4 * 16 * 2)
            pmc["frac_cycles_detail"] = "145 flop x executed RK4 steps / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CU x 4 SIMD x 16 f64 lanes x 2): clock-independent"
            pmc["sclk_MHz_in_profiled_pass"] = cycles / pmc["kernel_ns_in_write_pass"] * 1e3
        pmc_all[mode] = pmc
if pmc_all:
    json.dump(pmc_all, open(os.path.join(P, f"{R}_pmc_summary.json"), "w"), indent=1)
if summary:
    json.dump(summary, open(os.path.join(P, f"{R}_kernel_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

# render -> bloom -> sRGB8 (scripts/prof_rgb8.py): kernel stats + HBM bytes per kernel
src = find("prof_rgb8/**/rgb8_kernel_stats.csv")
if src:
    shutil.copy(src, os.path.join(P, f"{R}_rgb8_kernel_stats.csv"))
SHORT = ("copyBuffer", "fillBuffer", "trace_frame_kernel", "box_blur_sweep_rot", "box_blur_sweep_lds", "bloom_combine_srgb8", "srgb8_kernel")
rgb8 = collections.OrderedDict()
for grp, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    fn = find(f"pmc_{grp}_rgb8/**/{grp}_counter_collection.csv")
    if not fn:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == counter:
            for key in SHORT:
                if key in r["Kernel_Name"]:
                    agg[key].append(float(r["Counter_Value"]))
                    break
    for name, v in agg.items():
        rgb8.setdefault(name, {})[f"{counter}_KiB_per_launch"] = sum(v) / len(v)
for name, d in rgb8.items():
    if len(d) == 2:
        d["hbm_bytes_per_launch"] = 1024 * (2 * d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"])   # (the guide's gfx950 correction: FETCH_SIZE x 2)
if rgb8:
    json.dump(rgb8, open(os.path.join(P, f"{R}_rgb8_pmc_summary.json"), "w"), indent=1)

# the lines and notes of the pass, as they are
for fn in sorted(glob.glob(os.path.join(G, "bench_*.json"))) + [os.path.join(G, x) for x in ("host_topology.txt", "box.txt", "pytest_gpu_tail.txt", "png_files_vs_png_batch.jsonl", "smoke.txt")]:
    if os.path.exists(fn) and os.path.getsize(fn) > 2 and not os.path.basename(fn).startswith("bench_profiled"):
        shutil.copy(fn, os.path.join(P, f"{R}_{os.path.basename(fn)}"))
