#!/usr/bin/env python
"""bench.py -- headline benchmark: Mpixel/s of the geodesic trace on scenes/default-aa.yaml (BASELINE configs[2]).

One "step" = one pass of the hot path over one frame: 1920x1080 output pixels, 4x supersampled (8,294,400
traced rays), 470,000-star synthetic PPM-layout catalogue resident in HBM, image written to HBM.
N GPUs: one process per GPU (torchrun), every rank renders its own frames (frame-sharded, no data-path
collective); RCCL carries only the barrier and the max-over-ranks time (plus, with --gather, the final frames to rank 0).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 145  # 130 (rk4, src/Raytracer.hs:113-134) + 15 (findColor where-bindings, :100-102)
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X FP64 vector, FMA = 2 flop (= 1/2 of the guide's 157.3 TF FP32 vector peak)
PEAK_HBM_GBS = 8000.0


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
    return {"value": w * h / st["seconds"] / 1e6, "unit": "Mpixel/s", "cores": int(st["threads"]), "kind": "port",
            "rays_per_s": st["rays"] / st["seconds"], "seconds": st["seconds"],
            "sample": f"default-aa.yaml camera at {w}x{h} output px (4x supersampled = {st['rays']} rays), same 470k-star catalogue, "
                      f"C restatement of the reference CPU path (oracle/blackstar_oracle.c, -O2, pthreads over rows)"}


def pmc_traffic(mode):
    """HBM bytes per launch of the trace kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, KiB units; FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950 -- an upper
    bound here, since that calibration is for wide coalesced reads and this kernel's reads are 32-byte star-grid entries)."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(fn)).get(mode, {})
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, os.path.basename(fn)
        except (OSError, ValueError):
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["strict", "fast"], default=os.environ.get("BLACKSTAR_BENCH_MODE", "fast"))
    ap.add_argument("--workload", choices=["default-aa", "animation"], default="default-aa",
                    help="default-aa = BASELINE configs[2] (the headline metric); animation = configs[4]: frames of "
                         "animations/default-ani.yaml (nFrames overridden to 600), frame i on rank i %% N")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: also gather every rank's last frame to rank 0 inside the timed region (off by default: the path "
                         "shards by frame and has no exchange step; frames stay in the HBM of the GPU that rendered them, as at N=1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget (0 disables)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes/launch from separate rocprofv3 --pmc passes (default: read profiles/*_pmc_summary.json)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import blackstar_amd as bs
    from blackstar_amd import _lib, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # BLACKSTAR_BENCH_BACKEND=gloo lets the multi-process path be smoke-tested on a ONE-GPU box (ranks share device 0,
    # the gather goes through host memory); the real run is one rank per GPU over RCCL.
    backend = os.environ.get("BLACKSTAR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg_obj = bs.Config.from_file(os.path.join(ROOT, "scenes", "default-aa.yaml"))
    cfg = cfg_obj.to_bs_config()
    W, H = cfg["width"], cfg["height"]
    star_bytes = synthetic.ppm_catalogue_bytes()
    tree = bs.StarTree(bs.read_map(star_bytes), device=local_rank)
    tree.set_mode(_lib.BS_MODE_FAST if args.mode == "fast" else _lib.BS_MODE_STRICT)

    out = torch.empty((H, W, 3), dtype=torch.float64, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream()
    frames_cfg = None
    if args.workload == "animation":
        anim = bs.Animation.from_file(os.path.join(ROOT, "animations", "default-ani.yaml"))
        anim.nFrames = 600  # BASELINE configs[4] (the file itself says 375)
        bs.validate_keyframes(anim.keyframes)
        frames_cfg = [c.to_bs_config() for c in bs.generate_frames(anim)]
        cfg = frames_cfg[0]
        W, H = cfg["width"], cfg["height"]
    counter = {"i": 0}

    def step():
        c = cfg
        if frames_cfg is not None:  # frame i of the animation goes to rank i % world
            c = frames_cfg[(counter["i"] * world + rank) % len(frames_cfg)]
            counter["i"] += 1
        bs.render_device(c, tree, out.data_ptr(), out.numel(), stream.cuda_stream)

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
        a.record(stream)
        step()
        b.record(stream)
    if world > 1 and args.gather:
        gather_to_root()
    fence()
    dt = time.perf_counter() - t0
    st = tree.stats()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))  # per launch incl. the 64-B counter memset/copy nodes

    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())

    if rank == 0:
        frames = args.steps * world
        value = frames * W * H / dt / 1e6
        executed = int(st["steps"]) - int(st["rays"])  # the kernel skips the reference's final, discarded rk4 per ray
        flops = FLOP_PER_STEP * executed
        achieved = flops / (kernel_ms * 1e-3) / 1e12  # mean launch duration over the timed region (HIP events on the launch stream)
        alg_bytes = 24.0 * W * H
        traffic, traffic_src = (args.traffic_bytes, "--traffic-bytes") if args.traffic_bytes is not None else pmc_traffic(args.mode)
        res = {
            "metric": "Mpixel/s (geodesic rays/s) on default-aa.yaml", "value": value, "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("scenes/default-aa.yaml 1920x1080, 4x supersample (8,294,400 rays/frame), 470k-star synthetic "
                                    "PPM-layout catalogue, direction-grid star lookup (BASELINE configs[2])") if frames_cfg is None else
                                   ("animations/default-ani.yaml, nFrames=600, 1920x1080, 4x supersample, 470k-star synthetic catalogue, "
                                    "frame i on rank i % N (BASELINE configs[4]); roofline figures refer to the LAST frame rendered"),
                       "mode": args.mode, "frames_per_step_per_gpu": 1, "parallelism": f"frame-sharded x{world}",
                       "image": "RGB f64 resident in HBM (no D2H in the timed region)"},
            "rays_per_s": frames * st["rays"] / dt, "steps_per_ray": st["steps"] / st["rays"],
            "lane_efficiency": st["steps"] / (64.0 * st["wave_iters"]),
            "kernel_ms": kernel_ms, "kernel_ms_last_hipevent": st["kernel_ms"],
            "roofline": {"bound": "valu", "detail": "FP64 VALU issue (scalar ODE per lane; HBM and MFMA are not the bound)",
                         "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS,
                         "flop_per_launch": flops, "flop_per_step": FLOP_PER_STEP, "rk4_steps_executed": executed,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm": {"algorithmic_bytes": alg_bytes, "achieved_GBs": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                                 "peak_GBs": PEAK_HBM_GBS, "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}},
        }
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(cfg, star_bytes, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    tree.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
