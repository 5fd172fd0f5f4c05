#!/usr/bin/env python
"""bench.py -- headline benchmark: Mpixel/s of the geodesic trace on scenes/default-aa.yaml (BASELINE configs[2]).

One "step" = one pass of the hot path over one frame per GPU: 1920x1080 output pixels, 4x supersampled (8,294,400 traced
rays), 470,000-star synthetic PPM-layout catalogue resident in HBM, image written to HBM.  Frames are independent, so N GPUs
are frame-sharded with NO data-path collective.  Three ways to run N GPUs, all printing ONE JSON line:

  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   one process per GPU (launcher "torchrun-env");
                                                                             RCCL carries the barrier, the max-over-ranks
                                                                             time and, with --gather, the frames to rank 0
  python bench.py --gpus N                      no launcher around it -> ONE process, N bs_ctx (one per device), N streams
                                                (launcher "single-process": SURVEY.md 8e / app/Main.hs:68-77's batch loop
                                                with one context per GPU; no RCCL at all, --gather = peer copies)
  python bench.py --gpus N --launcher torchrun  re-executes itself under torch.distributed.run (first form)

On a box with fewer than N devices the single-process form still runs (contexts share devices round-robin, the JSON says
"oversubscribed": true) and the torchrun form falls back to gloo with ranks sharing devices -- smoke modes, not results.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 145  # 130 (rk4, src/Raytracer.hs:113-134) + 15 (findColor where-bindings, :100-102)
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X FP64 vector, FMA = 2 flop (= 1/2 of the guide's 157.3 TF FP32 vector peak)
PEAK_HBM_GBS = 8000.0
# VALU instructions the stepping loop issues per RK4 step of a wavefront (ISA count, scripts/isa_hot_blocks.py; static):
# full-rate f64 ops and quarter-rate transcendental seeds (v_rsq_f64 / v_rcp_f64 occupy the pipe for 4 issue slots).
LOOP_VALU = {"fast": {"full_rate": 62, "quarter_rate": 4}, "strict": {"full_rate": 178, "quarter_rate": 8}}
WORKLOAD_C3 = ("scenes/default-aa.yaml 1920x1080, 4x supersample (8,294,400 rays/frame), 470k-star synthetic "
               "PPM-layout catalogue, direction-grid star lookup (BASELINE configs[2])")
WORKLOAD_C5 = ("animations/default-ani.yaml, nFrames=600, 1920x1080, 4x supersample, 470k-star synthetic catalogue, "
               "frame i on rank i % N (BASELINE configs[4]); roofline figures refer to the LAST frame rendered")


def cpu_baseline(cfg, star_bytes, budget_s):
    """Time the C oracle (restatement of the reference CPU path; GHC is unavailable) on a bounded sample."""
    from oracle import c_oracle, scenes
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container CPU quota (cgroup v2 cpu.max = "<quota> <period>") caps the cores that really run
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    ix = c_oracle.Index(c_oracle.read_ppm(star_bytes))
    probe = scenes.with_res(cfg, 96, 54)
    _, st = c_oracle.render(probe, ix, threads=threads)
    rate = st["rays"] / max(st["seconds"], 1e-6)  # rays/s
    rays = min(rate * budget_s, 4.0 * cfg["width"] * cfg["height"])
    scale = (rays / (4.0 * cfg["width"] * cfg["height"])) ** 0.5
    w = max(16, int(cfg["width"] * scale) // 16 * 16)
    h = max(9, w * cfg["height"] // cfg["width"])
    sample = scenes.with_res(cfg, w, h)
    _, st = c_oracle.render(sample, ix, threads=threads)
    # BASELINE configs[0] (the reference's own CPU-runnable case), whole: default.yaml at 640x480, no supersampling, no star map
    _, st1 = c_oracle.render(scenes.with_res(scenes.DEFAULT, 640, 480), c_oracle.Index(None), threads=threads)
    c1 = {"value": 640 * 480 / st1["seconds"] / 1e6, "unit": "Mpixel/s", "seconds": st1["seconds"], "rays": int(st1["rays"]),
          "config": "scenes/default.yaml 640x480, no supersampling, no star map (BASELINE configs[0]), the whole frame"}
    return {"value": w * h / st["seconds"] / 1e6, "unit": "Mpixel/s", "cores": int(st["threads"]), "kind": "port", "configs0": c1,
            "rays_per_s": st["rays"] / st["seconds"], "seconds": st["seconds"],
            "sample": f"default-aa.yaml camera at {w}x{h} output px (4x supersampled = {st['rays']} rays), same 470k-star catalogue, "
                      f"C restatement of the reference CPU path (oracle/blackstar_oracle.c, -O2, pthreads over rows)"}


def pmc_traffic(mode):
    """HBM bytes per launch of the trace kernel from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, KiB units; FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950 -- an upper
    bound here, since that calibration is for wide coalesced reads and this kernel's reads are 32-byte star-grid entries).
    A static figure: counters cannot be collected from inside an un-profiled run."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(fn)).get(mode, {})
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, os.path.basename(fn)
        except (OSError, ValueError):
            pass
    return None, None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["strict", "fast"], default=os.environ.get("BLACKSTAR_BENCH_MODE", "fast"))
    ap.add_argument("--workload", choices=["default-aa", "animation"], default="default-aa",
                    help="default-aa = BASELINE configs[2] (the headline metric); animation = configs[4]: frames of "
                         "animations/default-ani.yaml (nFrames overridden to 600), frame i on rank i %% N")
    ap.add_argument("--launcher", choices=["auto", "single-process", "torchrun"], default="auto",
                    help="how --gpus N > 1 runs when no torch.distributed launcher started this process "
                         "(auto = single-process: N contexts, one per device, in this process)")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: also gather every rank's last frame to rank 0 inside the timed region (off by default: the path "
                         "shards by frame and has no exchange step; frames stay in the HBM of the GPU that rendered them, as at N=1)")
    ap.add_argument("--streams", type=int, choices=[1, 2], default=None,
                    help="launches in flight per GPU: consecutive frames alternate between this many streams (and output images). "
                         "Default 1 for default-aa (per-launch event times, rocprofv3 kernel durations and ms_per_step stay one number), "
                         "2 for the animation workload (independent frames: the next frame's launch fills the SIMDs this frame's last tiles leave)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget (0 disables)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the bs_render / bs_render_rgb8 / STRICT / ubench legs at N=1")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes/launch from separate rocprofv3 --pmc passes (default: read profiles/*_pmc_summary.json)")
    return ap.parse_args()


def load_workload(args, bs):
    cfg_obj = bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml"))
    cfg = cfg_obj.to_bs_config()
    frames_cfg = None
    if args.workload == "animation":
        anim = bs.Animation.from_file(os.path.join(ROOT, "animations", "default-ani.yaml"))
        anim.nFrames = 600  # BASELINE configs[4] (the file itself says 375)
        bs.validate_keyframes(anim.keyframes)
        frames_cfg = [c.to_bs_config() for c in bs.generate_frames(anim)]
        cfg = frames_cfg[0]
    return cfg_obj, cfg, frames_cfg


def roofline_block(args, st, kernel_ms, W, H, peak_measured=None):
    executed = int(st["steps"]) - int(st["rays"])  # the kernel skips the reference's final, discarded rk4 per ray
    flops = FLOP_PER_STEP * executed
    achieved = flops / (kernel_ms * 1e-3) / 1e12  # mean launch duration over the timed region (HIP events on the launch stream)
    alg_bytes = 24.0 * W * H
    traffic, traffic_src = (args.traffic_bytes, "--traffic-bytes") if args.traffic_bytes is not None else pmc_traffic(args.mode)
    r = {"bound": "valu", "detail": "FP64 VALU issue (scalar ODE per lane; HBM and MFMA are not the bound)",
         "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS,
         "flop_kind": "reference-equivalent: 145 flop per RK4 step as the reference's arithmetic counts them (SURVEY 8d), "
                      "NOT executed instructions -- see valu_issue_frac for those",
         "flop_per_launch": flops, "flop_per_step": FLOP_PER_STEP, "rk4_steps_executed": executed,
         "traffic": traffic, "traffic_kind": "static (committed rocprofv3 --pmc passes, not measured in this run)" if args.traffic_bytes is None else "given",
         "traffic_source": traffic_src,
         "hbm": {"algorithmic_bytes": alg_bytes, "achieved_GBs": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                 "peak_GBs": PEAK_HBM_GBS, "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}}
    if peak_measured:
        lv = LOOP_VALU[args.mode]
        # issue slots the stepping loop needs: every wavefront iteration issues full_rate + 4 * quarter_rate slots of 64 lanes
        slots = float(st["wave_iters"]) * 64.0 * (lv["full_rate"] + 4 * lv["quarter_rate"])
        r["peak_measured"] = peak_measured["TFLOPs"]
        r["peak_measured_detail"] = peak_measured["detail"]
        r["frac_of_measured_peak"] = achieved / peak_measured["TFLOPs"]
        r["valu_issue_frac"] = slots / (kernel_ms * 1e-3) / (peak_measured["Ginstr_per_s"] * 1e9)
        r["valu_issue_detail"] = (f"stepping-loop VALU issue slots ({lv['full_rate']} full-rate f64 + {lv['quarter_rate']} quarter-rate per "
                                  "wavefront step, ISA count) x wavefront iterations / launch time, over the v_fma_f64 issue rate measured in this run")
    return r


def measure_peak(tree, _lib):
    """v_fma_f64 issue rate on this box, this run (bs_debug_ubench: 8 independent chains per lane, 2048 workgroups)."""
    import ctypes as C
    L = _lib.lib()
    ms, gi = C.c_double(), C.c_double()
    best = 0.0
    for _ in range(3):
        _lib.check(L.bs_debug_ubench(tree.handle, 0, 256 * 8, 20000, C.byref(ms), C.byref(gi)), "bs_debug_ubench")
        best = max(best, gi.value / ms.value * 1e3)
    return {"Ginstr_per_s": best, "TFLOPs": best * 2 / 1e3, "detail": "v_fma_f64, 8 chains/lane, best of 3 (bs_debug_ubench)"}


def boundary_numbers(bs, _lib, tree, cfg_obj, cfg, args, torch, out, stream):
    """What SURVEY 8d asks for beside the kernel-only figure: the wall time of the drop-in calls themselves (kernel + D2H),
    and the STRICT mode of the same frame.  Runs after the timed loop, outside `value`."""
    import numpy as np
    W, H = cfg["width"], cfg["height"]
    res = {}

    def med(f, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    def entry(ms, note):
        return {"ms": ms, "Mpixel_s": W * H / ms / 1e3, "note": note}

    t0 = time.perf_counter()
    pinned = bs.alloc_image(tree, H, W)
    res["bs_host_alloc_ms"] = (time.perf_counter() - t0) * 1e3  # why the shim allocates its page-locked image buffer ONCE
    bs.render(cfg, tree, out=pinned)
    res["bs_render_pinned"] = entry(med(lambda: bs.render(cfg, tree, out=pinned), 5),
                                    "bs_render into a bs_host_alloc (page-locked) buffer: the kernel writes the frame straight into host memory over PCIe (zero copy), blocking")
    touched = np.empty((H, W, 3))
    bs.render(cfg, tree, out=touched)
    res["bs_render_pageable_reused"] = entry(med(lambda: bs.render(cfg, tree, out=touched), 5),
                                             "bs_render into ONE pageable buffer reused across calls: kernel (two half-frame launches) + 49.8 MB staged D2H")
    res["bs_render_pageable"] = entry(med(lambda: bs.render(cfg, tree, out=np.empty((H, W, 3))), 3),
                                      "bs_render into a freshly allocated pageable buffer every call (first-touch page faults included)")
    pinned8 = bs.alloc_image(tree, H, W, dtype=np.uint8)
    bs.render_rgb8(cfg_obj, tree, out=pinned8)
    res["bs_render_rgb8"] = entry(med(lambda: bs.render_rgb8(cfg_obj, tree, out=pinned8), 5),
                                  "render + bloom + sRGB8 on the device, 6.2 MB RGB8 written into a page-locked host buffer by the last kernel "
                                  "(doRender up to the PNG encoder), blocking")
    # Two frames in flight: consecutive frames alternate between two streams (what bs_render_batch does per context), so the end of
    # one launch -- the ~0.3 ms in which its last tiles drain and the SIMDs empty -- overlaps the start of the next
    out2 = torch.empty_like(out)
    s2 = torch.cuda.Stream()
    lanes = [(out, stream), (out2, s2)]
    n2 = 20
    for k in range(4):
        o, s = lanes[k & 1]
        bs.render_device(cfg, tree, o.data_ptr(), o.numel(), s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n2):
        o, s = lanes[k & 1]
        bs.render_device(cfg, tree, o.data_ptr(), o.numel(), s.cuda_stream)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) * 1e3 / n2
    res["two_streams"] = entry(ms2, f"{n2} frames resident in HBM, alternating between two streams (two launches in flight): wall time per frame")
    # STRICT mode of the same frame, image resident in HBM like the headline
    tree.set_mode(_lib.BS_MODE_STRICT)
    try:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        bs.render_device(cfg, tree, out.data_ptr(), out.numel(), stream.cuda_stream)
        for a, b in ev:
            a.record(stream)
            bs.render_device(cfg, tree, out.data_ptr(), out.numel(), stream.cuda_stream)
            b.record(stream)
        torch.cuda.synchronize()
        st = tree.stats()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        executed = int(st["steps"]) - int(st["rays"])
        tf = FLOP_PER_STEP * executed / (ms * 1e-3) / 1e12
        strict = {"ms_per_step": ms, "Mpixel_s": W * H / ms / 1e3, "achieved_TFLOPs": tf, "frac": tf / PEAK_FP64_VALU_TFLOPS,
                  "note": "BS_MODE_STRICT (bit-exact trajectories), image resident in HBM, 4 launches"}
    finally:
        tree.set_mode(_lib.BS_MODE_FAST if args.mode == "fast" else _lib.BS_MODE_STRICT)
    return res, strict


def result_line(args, world, launcher, value, dt, W, H, frames_cfg, st, kernel_ms, extra_cfg=None, peak_measured=None):
    overlapped = bool(extra_cfg) and extra_cfg.get("launches_in_flight_per_gpu", 1) > 1
    cfgd = {"workload": WORKLOAD_C3 if frames_cfg is None else WORKLOAD_C5, "mode": args.mode, "frames_per_step_per_gpu": 1,
            "parallelism": f"frame-sharded x{world}", "launcher": launcher,
            "image": "RGB f64 resident in HBM (no D2H in the timed region)"}
    if extra_cfg:
        cfgd.update(extra_cfg)
    frames = args.steps * world
    return {
        "metric": "Mpixel/s (geodesic rays/s) on default-aa.yaml", "value": value, "unit": "Mpixel/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": cfgd,
        "rays_per_s": frames * st["rays"] / dt, "steps_per_ray": st["steps"] / st["rays"],
        "lane_efficiency": st["steps"] / (64.0 * st["wave_iters"]),
        "kernel_ms": kernel_ms, "kernel_ms_last_hipevent": st["kernel_ms"],
        # launches in flight overlap: a launch's own duration then says nothing about the rate; the step time does
        "roofline": dict(roofline_block(args, st, kernel_ms if not overlapped else dt / args.steps * 1e3, W, H, peak_measured),
                         time_basis="mean launch duration (HIP events)" if not overlapped else
                         "ms_per_step (launches overlap: two in flight per GPU; their own durations are about twice this)"),
    }


def run_ranks(args):
    """One process per GPU (this process is one rank; torch.distributed.run or the driver's launcher set the env)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import blackstar_amd as bs
    from blackstar_amd import _lib, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # BLACKSTAR_BENCH_BACKEND=gloo lets the multi-process path be smoke-tested on a box with fewer GPUs than ranks (ranks
    # share devices, the gather goes through host memory); the real run is one rank per GPU over RCCL.
    backend = os.environ.get("BLACKSTAR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg_obj, cfg, frames_cfg = load_workload(args, bs)
    W, H = cfg["width"], cfg["height"]
    star_bytes = synthetic.ppm_catalogue_bytes()
    tree = bs.StarTree(bs.read_map(star_bytes), device=local_rank)
    tree.set_mode(_lib.BS_MODE_FAST if args.mode == "fast" else _lib.BS_MODE_STRICT)

    out = torch.empty((H, W, 3), dtype=torch.float64, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream()
    n_streams = args.streams or (2 if frames_cfg is not None else 1)
    lanes = [(out, stream)] + [(torch.empty_like(out), torch.cuda.Stream()) for _ in range(n_streams - 1)]
    counter = {"i": 0, "k": 0}

    def step():
        c = cfg
        if frames_cfg is not None:  # frame i of the animation goes to rank i % world
            c = frames_cfg[(counter["i"] * world + rank) % len(frames_cfg)]
            counter["i"] += 1
        o, s = lanes[counter["k"] % n_streams]
        counter["k"] += 1
        bs.render_device(c, tree, o.data_ptr(), o.numel(), s.cuda_stream)
        return s

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_to_root():  # optional (--gather): each rank's finished frame goes to rank 0 over xGMI
        src = out if backend == "nccl" else out.cpu()
        gathered = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
        dist.gather(src, gathered, dst=0)
        return gathered

    for _ in range(args.warmup):
        step()
    if world > 1 and args.gather:
        gather_to_root()  # also establishes RCCL's point-to-point channels outside the timed region
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        s = lanes[counter["k"] % n_streams][1]
        a.record(s)
        step()
        b.record(s)
    t_gather = None
    if world > 1 and args.gather:
        torch.cuda.synchronize()
        tg = time.perf_counter()
        gather_to_root()
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - tg
    fence()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    st = tree.stats()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))  # per launch incl. the 64-B counter memset/copy nodes

    per_rank_ms = [dt_local / args.steps * 1e3]
    if world > 1:
        dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        tdt = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        allt = [torch.empty_like(tdt) for _ in range(world)]
        dist.all_gather(allt, tdt)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in allt]
        dt = max(float(t.item()) for t in allt)  # MAX over ranks

    if rank == 0:
        frames = args.steps * world
        value = frames * W * H / dt / 1e6
        peak = None
        if world == 1 and not args.no_boundary and frames_cfg is None:
            peak = measure_peak(tree, _lib)
        extra = {"backend": ("RCCL (nccl)" if backend == "nccl" else backend) if world > 1 else "none (single rank)",
                 "devices_visible": ndev, "oversubscribed": world > ndev, "launches_in_flight_per_gpu": n_streams}
        if n_streams > 1:
            extra["launches_in_flight_note"] = ("consecutive frames alternate between two streams and share the GPU, so kernel_ms "
                                                "(per-launch event time) exceeds ms_per_step")
        res = result_line(args, world, "torchrun-env (one process per GPU)" if world > 1 or "WORLD_SIZE" in os.environ else "single-process",
                          value, dt, W, H, frames_cfg, st, kernel_ms, extra, peak)
        res["per_rank_ms_per_step"] = per_rank_ms
        if t_gather is not None:
            res["gather_ms"] = t_gather * 1e3
            res["config"]["gather"] = "dist.gather of every rank's last frame to rank 0 inside the timed region"
        if world == 1 and not args.no_boundary and frames_cfg is None:
            res["boundary"], res["strict"] = boundary_numbers(bs, _lib, tree, cfg_obj, cfg, args, torch, out, stream)
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(cfg, star_bytes, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    tree.close()
    if world > 1:
        dist.destroy_process_group()


def run_single_process(args):
    """N GPUs from ONE process: one bs_ctx + one output image + one stream per device, every step enqueues one frame on each
    (bs_render_device is asynchronous, so one host thread keeps N GPUs busy); no collective of any kind."""
    import numpy as np
    import torch

    import blackstar_amd as bs
    from blackstar_amd import _lib, synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    world = args.gpus
    ndev = torch.cuda.device_count()
    devs = [i % ndev for i in range(world)]
    cfg_obj, cfg, frames_cfg = load_workload(args, bs)
    W, H = cfg["width"], cfg["height"]
    star_bytes = synthetic.ppm_catalogue_bytes()
    stars = bs.read_map(star_bytes)
    n_streams = args.streams or (2 if frames_cfg is not None else 1)
    trees, outs, streams, lanes = [], [], [], []
    for d in devs:
        t = bs.StarTree(stars, device=d)
        t.set_mode(_lib.BS_MODE_FAST if args.mode == "fast" else _lib.BS_MODE_STRICT)
        trees.append(t)
        with torch.cuda.device(d):
            ln = [(torch.empty((H, W, 3), dtype=torch.float64, device=f"cuda:{d}"), torch.cuda.Stream(device=d)) for _ in range(n_streams)]
        lanes.append(ln)
        outs.append(ln[0][0])
        streams.append(ln[0][1])
    counter = {"i": 0, "k": [0] * world}

    def lane(k):  # the (image, stream) device k's NEXT frame goes to
        return lanes[k][counter["k"][k] % n_streams]

    def step(k):
        c = cfg
        if frames_cfg is not None:
            c = frames_cfg[(counter["i"] * world + k) % len(frames_cfg)]
        o, st_ = lane(k)
        counter["k"][k] += 1
        bs.render_device(c, trees[k], o.data_ptr(), o.numel(), st_.cuda_stream)

    def fence():
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    gathered = None
    if args.gather:
        with torch.cuda.device(devs[0]):
            gathered = [torch.empty_like(outs[0]) for _ in range(world)]

    def gather_to_root():  # peer copies into device devs[0] (hipMemcpyPeerAsync over xGMI), each on its source stream
        for k in range(world):
            with torch.cuda.device(devs[k]), torch.cuda.stream(streams[k]):
                gathered[k].copy_(outs[k], non_blocking=True)

    for _ in range(args.warmup):
        for k in range(world):
            step(k)
        counter["i"] += 1
    if args.gather:
        gather_to_root()
    fence()
    ev = []
    for k in range(world):
        with torch.cuda.device(devs[k]):
            ev.append([(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)])
    t0 = time.perf_counter()
    for s in range(args.steps):
        for k in range(world):
            with torch.cuda.device(devs[k]):
                st_ = lane(k)[1]
                ev[k][s][0].record(st_)
                step(k)
                ev[k][s][1].record(st_)
        counter["i"] += 1
    t_gather = None
    if args.gather:
        fence()
        tg = time.perf_counter()
        gather_to_root()
        fence()
        t_gather = time.perf_counter() - tg
    fence()
    dt = time.perf_counter() - t0  # one clock for all devices: this IS the max over "ranks"
    st = trees[0].stats()
    kms = [[a.elapsed_time(b) for a, b in ev[k]] for k in range(world)]
    per_rank_ms = [dt / args.steps * 1e3] * world if n_streams > 1 else [ev[k][0][0].elapsed_time(ev[k][-1][1]) / args.steps for k in range(world)]
    kernel_ms = float(np.mean(kms[0]))
    value = args.steps * world * W * H / dt / 1e6
    extra = {"backend": "none (one process, one bs_ctx + stream per device; frames never leave their GPU)",
             "devices_visible": ndev, "oversubscribed": world > ndev, "devices": devs, "launches_in_flight_per_gpu": n_streams}
    res = result_line(args, world, "single-process (N contexts)", value, dt, W, H, frames_cfg, st, kernel_ms, extra)
    res["per_rank_ms_per_step"] = per_rank_ms
    res["per_rank_kernel_ms"] = [float(np.mean(x)) for x in kms]
    if t_gather is not None:
        res["gather_ms"] = t_gather * 1e3
        res["config"]["gather"] = "peer copy of every device's last frame to device 0 inside the timed region"
    print(json.dumps(res), flush=True)
    for t in trees:
        t.close()


def reexec_under_torchrun(args):
    """`python bench.py --gpus N --launcher torchrun`: become the launcher of the one-process-per-GPU form."""
    import torch
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
        env.setdefault("BLACKSTAR_BENCH_BACKEND", "gloo")  # smoke mode: ranks share devices, RCCL needs one device per rank
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if "WORLD_SIZE" in os.environ or args.gpus == 1:
        run_ranks(args)
    elif args.launcher == "torchrun":
        reexec_under_torchrun(args)
    else:
        run_single_process(args)


if __name__ == "__main__":
    main()
