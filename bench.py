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

`value` is always the RESIDENT form (frames stay in HBM, the contract's "inputs already resident" figure).  Beside it the same
line carries, measured after the timed region (--form all, the default):
  with_d2h.batch       bs_render_batch: the product's own multi-frame / multi-GPU entry point (one host thread per context, two
                       frames in flight per context), every frame delivered as RGB f64 into page-locked host memory (zero copy)
  with_d2h.rgb8_batch  bs_render_rgb8_batch: the reference's batch loop over doRender (app/Main.hs:68-77, :105-123) -- render,
                       bloom, sRGB8 on the device, only RGB8 leaves the GPU
  with_d2h.png_batch   bs_render_png_batch: the same with writeImg's PNG encoder on the device too -- the finished FILE leaves the GPU
  with_d2h.png_files   bs_render_png_files: ... and is written to a RAM-disk file by the library's writer thread (scene to file)
  with_d2h.split       ONE frame of BASELINE configs[3] (lensing-disk at 3840x2160) cut into row bands over all GPUs (bs_render_split, or
                       bs_render_rows per rank), compared byte for byte with one device's frame; speedup_vs_one_device = strong scaling
  per_config           N = 1: BASELINE configs[1] (default.yaml, no star map) and configs[3] (lensing-disk at 4K) -- hipEvent ms, Mpixel/s,
                       steps, roofline frac of three launches each; --workload default | lensing-4k makes either the timed workload
  validation / valid   after the timed region every device renders the same frame once more: the frames must be BIT-IDENTICAL across
                       devices (sha256) and the step counters equal; every delivered form compares its frames too (frames_identical).
                       A mismatch makes the line "valid": false -- a speed-up is only a result if the other GPUs rendered the scene
  sustained            300 more frames on one stream with per-50-frame times and sampled sclk / power (clock droop under the
                       package power cap is visible here, not in 20 launches)
--catalogue clustered | PATH swaps the uniform synthetic sky for the non-uniform one or a real PPM catalogue file (reported as such).
Which key answers BASELINE's ">= 6x at 8 GPUs": for frames (configs[4], and the headline) value(N=8) / value(N=1) of the driver's own runs
-- weak scaling, "valid" must be true; for one huge frame with_d2h.split.speedup_vs_one_device of the N = 8 line (or --form split).
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

T_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs import *  # noqa: E402,F401,F403  (the legs, the workloads table and the result line: bench_legs.py)
from bench_legs import CATALOGUES, COUNTERS, WORKLOADS  # noqa: E402,F401  (named for the readers of this file)


PARITY_TOLERANCE = "|gpu - cpu| <= 1e-4 * |cpu| + 1e-7 per channel per output pixel (SURVEY.md 8d 'Parity check'; north_star: 1e-4 relative)"
PARITY_COUNTERS = ("rays", "steps", "capped", "horizon", "escaped", "disk_hits", "star_hits")
# output sizes at which the oracle renders BASELINE configs[3] (lensing-disk at 3840x2160) and three frames of configs[4] (the animation at
# 1920x1080) for the parity list: about a second of the host's cores each; both 4x supersampled like the configs themselves
PARITY_C4_RES, PARITY_C5_RES, PARITY_C5_FRAMES = (640, 360), (480, 270), (0, 300, 599)


def parity_block(np, what, ref, ref_st, got, got_st, mode):
    """One config's frame from the HIP library (got) against the CPU oracle's frame of the SAME config and catalogue (ref): the numbers
    SURVEY.md 8d / BASELINE.md section 3 ask for beside the timing.  The oracle is the checker here, never the thing measured."""
    ref = np.asarray(ref, np.float64)
    got = np.asarray(got, np.float64)
    if ref.shape != got.shape:
        return {"config": what, "mode": mode, "error": f"shapes differ: oracle {ref.shape}, gpu {got.shape}", "outside_1e-4": int(ref.size)}
    diff = np.abs(got - ref)
    finite = np.isfinite(got)
    bad = ~(diff <= 1e-4 * np.abs(ref) + 1e-7)   # (a NaN on either side counts as outside)
    big = np.abs(ref) > 1e-3
    counters = {k: [int(ref_st[k]), int(got_st[k])] for k in PARITY_COUNTERS}
    blk = {"config": what, "mode": mode, "values": int(ref.size), "outside_1e-4": int(bad.sum()), "nonfinite": int((~finite).sum()),
           "max_abs": float(np.nanmax(diff)) if diff.size else 0.0,
           "max_rel_where_ref>1e-3": float(np.nanmax(diff[big] / np.abs(ref[big]))) if big.any() else 0.0,
           "bit_identical": bool(np.array_equal(ref, got)),
           "steps_equal": counters["steps"][0] == counters["steps"][1] and counters["rays"][0] == counters["rays"][1],
           "fates_equal": all(counters[k][0] == counters[k][1] for k in ("capped", "horizon", "escaped", "disk_hits", "star_hits")),
           "counters_oracle_gpu": counters}
    if blk["outside_1e-4"]:
        ys, xs, cs = np.nonzero(bad)
        blk["first_outside"] = [{"y": int(y), "x": int(x), "channel": int(c), "oracle": float(ref[y, x, c]), "gpu": float(got[y, x, c])}
                                for y, x, c in list(zip(ys, xs, cs))[:4]]
    return blk


def cpu_baseline(cfg, star_bytes, budget_s, gpu_render=None, np=None, baseline_config="configs[2]", host_cpus=None):
    """Time the C oracle (restatement of the reference CPU path; GHC is unavailable) on a bounded sample -- and, since the oracle's
    frames are computed anyway, compare them with the frames the HIP library renders of the same configs (gpu_render(cfg, stars, mode)
    -> (image, stats), the product called through its C ABI): the `parity` list.  No extra oracle time.
    host_cpus: the CPUs this process could use BEFORE it bound itself to its GPU's NUMA node (bind_rank_to_gpu_node): the CPU baseline is
    "the host's cores", all of them -- the oracle's threads run on that set, and the caller's binding comes back afterwards."""
    bound_to = None
    if host_cpus and hasattr(os, "sched_setaffinity"):
        try:
            bound_to = os.sched_getaffinity(0)
            os.sched_setaffinity(0, host_cpus)   # (threads the oracle starts inherit the calling thread's affinity)
        except OSError:
            bound_to = None
    try:
        return _cpu_baseline(cfg, star_bytes, budget_s, gpu_render, np, baseline_config)
    finally:
        if bound_to is not None:
            try:
                os.sched_setaffinity(0, bound_to)
            except OSError:
                pass


def _cpu_baseline(cfg, star_bytes, budget_s, gpu_render, np, baseline_config):
    from oracle import c_oracle, scenes
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container CPU quota (cgroup v2 cpu.max = "<quota> <period>") caps the cores that really run
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    ix = c_oracle.Index(c_oracle.read_ppm(star_bytes) if star_bytes else None)
    probe = scenes.with_res(cfg, 96, 54)
    _, st = c_oracle.render(probe, ix, threads=threads)
    rate = st["rays"] / max(st["seconds"], 1e-6)  # rays/s
    ss = 4.0 if cfg["supersampling"] else 1.0
    rays = min(rate * budget_s, ss * cfg["width"] * cfg["height"])
    scale = (rays / (ss * cfg["width"] * cfg["height"])) ** 0.5
    w = max(16, int(cfg["width"] * scale) // 16 * 16)
    h = max(9, w * cfg["height"] // cfg["width"])
    sample = scenes.with_res(cfg, w, h)
    img, st = c_oracle.render(sample, ix, threads=threads)
    # BASELINE configs[0] (the reference's own CPU-runnable case), whole: default.yaml at 640x480, no supersampling, no star map
    cfg1 = scenes.with_res(scenes.DEFAULT, 640, 480)
    img1, st1 = c_oracle.render(cfg1, c_oracle.Index(None), threads=threads)
    c1 = {"value": 640 * 480 / st1["seconds"] / 1e6, "unit": "Mpixel/s", "seconds": st1["seconds"], "rays": int(st1["rays"]),
          "config": "scenes/default.yaml 640x480, no supersampling, no star map (BASELINE configs[0]), the whole frame"}
    # BASELINE configs[1], whole: default.yaml 1920x1080, no supersampling, no star map (SURVEY 8d asks for C1, C2 and C3)
    img2, st2 = c_oracle.render(scenes.DEFAULT, c_oracle.Index(None), threads=threads)
    c2 = {"value": 1920 * 1080 / st2["seconds"] / 1e6, "unit": "Mpixel/s", "seconds": st2["seconds"], "rays": int(st2["rays"]),
          "config": "scenes/default.yaml 1920x1080, no supersampling, no star map (BASELINE configs[1]), the whole frame"}
    res = {"value": w * h / st["seconds"] / 1e6, "unit": "Mpixel/s", "cores": int(st["threads"]), "kind": "port", "configs0": c1, "configs1": c2,
           "rays_per_s": st["rays"] / st["seconds"], "seconds": st["seconds"],
           "sample": f"the workload's camera and scene at {w}x{h} output px ({'4x supersampled = ' if ss > 1 else ''}{st['rays']} rays), "
                     f"{'same %d-star catalogue' % len(ix.stars) if star_bytes else 'no star map'}, "
                     f"C restatement of the reference CPU path (oracle/blackstar_oracle.c, -O2, pthreads over rows)"}
    if gpu_render is not None:
        whole = (w, h) == (cfg["width"], cfg["height"])
        # (what, baseline_config, config, with the catalogue?, the oracle's frame and statistics -- None: render it now --, arithmetic modes)
        jobs = [("the timed workload" + ("" if whole else f" at the CPU sample's resolution {w}x{h}") + " (camera, scene and catalogue of the timed workload)"
                 if star_bytes else "the timed workload" + ("" if whole else f" at {w}x{h}"), baseline_config, sample, bool(star_bytes), img, st, ("fast",)),
                ("scenes/default.yaml 1920x1080, no supersampling, no star map, the whole frame", "configs[1]", dict(scenes.DEFAULT), False, img2, st2,
                 ("fast", "strict")),
                ("scenes/default.yaml 640x480, no supersampling, no star map, the whole frame", "configs[0]", cfg1, False, img1, st1, ("fast",))]
        if star_bytes:   # (the remaining BASELINE configs need the catalogue: SURVEY.md 8d "Parity check ... for each config")
            # configs[3]: lensing-disk.yaml, 4x supersampled, down-scaled from 3840x2160 to what the oracle renders in about a second
            w4, h4 = PARITY_C4_RES
            jobs.append((f"scenes/lensing-disk.yaml at {w4}x{h4} (BASELINE's 3840x2160 down-scaled for the oracle), 4x supersample, same catalogue", "configs[3]",
                         scenes.with_res(scenes.LENSING_DISK, w4, h4), True, None, None, ("fast",)))
            # configs[4]: frames 0, 300 and 599 of the 600-frame animation, 4x supersampled, at 480x270
            w5, h5 = PARITY_C5_RES
            for i in PARITY_C5_FRAMES:
                jobs.append((f"animations/default-ani.yaml, nFrames=600: frame {i} at {w5}x{h5} (1920x1080 down-scaled for the oracle), 4x supersample, same catalogue",
                             "configs[4]", scenes.with_res(scenes.ani_frame(i, 600), w5, h5), True, None, None, ("fast",)))
        res["parity"] = []
        for what, which, c, stars_, ref, ref_st, modes in jobs:
            for mode in modes:
                try:
                    if ref is None:
                        ref, ref_st = c_oracle.render(c, ix, threads=threads)
                    got, got_st = gpu_render(c, stars_, mode)
                    res["parity"].append(dict(parity_block(np, what, ref, ref_st, got, got_st, mode), baseline_config=which))
                except Exception as e:  # a leg that cannot run is reported as failing parity, not dropped
                    res["parity"].append({"config": what, "baseline_config": which, "mode": mode, "error": f"{type(e).__name__}: {e}", "outside_1e-4": -1})
        res["parity_configs"] = sorted({p_["baseline_config"] for p_ in res["parity"]})
        res["parity_tolerance"] = PARITY_TOLERANCE
        res["parity_ok"] = all(p.get("outside_1e-4") == 0 and p.get("steps_equal") and p.get("fates_equal") for p in res["parity"])
    return res


_RESULT_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout -- but native libraries write there too (RCCL's version banner, gloo's "connected to N peer
    ranks", profiler notes).  From here on file descriptor 1 is stderr's for everybody in this process; the result line alone goes to the
    real stdout, kept aside in _RESULT_FD (emit)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(res):
    line = (json.dumps(res) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
        return
    while line:
        line = line[os.write(_RESULT_FD, line):]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps before the timed ones.  The chip comes out of idle while the host builds the catalogue and the star grid: "
                         "the first launches after an idle spell run 5.3, 5.0, 4.8, 4.65, 4.5, 4.4 ms before the clocks settle at 4.3 (kernel_ms_each), "
                         "so the default gives that ramp ten launches")
    ap.add_argument("--warmup-ms", type=float, default=None,
                    help="keep warming up beyond --warmup until this many ms of launches have run (one process per GPU, resident form).  Default: 40 when "
                         "--warmup is not given -- ten launches of the headline frame are 43 ms, ten of a 1.2 ms frame (--workload default) are not, and "
                         "its timed launches would ride the clock ramp -- and 0 (exactly --warmup steps) when it is")
    ap.add_argument("--mode", choices=["strict", "fast"], default=os.environ.get("BLACKSTAR_BENCH_MODE", "fast"))
    ap.add_argument("--workload", choices=["default-aa", "default", "lensing-4k", "animation"], default="default-aa",
                    help="default-aa = BASELINE configs[2] (the headline metric); default = configs[1]: scenes/default.yaml 1920x1080, no "
                         "supersampling, no star map; lensing-4k = configs[3]: scenes/lensing-disk.yaml at 3840x2160, 4x supersample; "
                         "animation = configs[4]: frames of animations/default-ani.yaml (nFrames overridden to 600), frame i on rank i %% N")
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
    ap.add_argument("--catalogue", default="synthetic",
                    help="synthetic (uniform 470k-star sky, the BASELINE input) | clustered (non-uniform: + clusters + a dense band) | "
                         "PATH of a real PPM catalogue file in the layout src/StarMap.hs:45-58 reads (reported separately)")
    ap.add_argument("--form", choices=["all", "resident", "batch", "rgb8-batch", "png-batch", "png-files", "split"], default="all",
                    help="`value` is the resident form (image stays in HBM) unless batch / rgb8-batch / png-batch / png-files / split is named here; "
                         "all (default) = resident as `value` plus the with_d2h block (bs_render_batch, bs_render_rgb8_batch, bs_render_png_batch, "
                         "bs_render_png_files into page-locked host memory, and `split`).  split = ONE frame of BASELINE configs[3] (lensing-disk at "
                         "3840x2160) cut into row bands over all N GPUs (bs_render_split / bs_render_rows, SURVEY 8e's fallback for a single huge "
                         "frame): total work is fixed, so the line says \"scaling\": \"strong\"")
    ap.add_argument("--no-validate", action="store_true",
                    help="skip the untimed validation after the timed region (every device renders the same frame once more; the frames must be "
                         "bit-identical and the step counters equal)")
    ap.add_argument("--sustained-frames", type=int, default=300, help="frames of the `sustained` leg after the timed region (0 disables)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget (0 disables)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the bs_render / bs_render_rgb8 / STRICT / ubench legs at N=1")
    ap.add_argument("--traffic-bytes", type=float, default=None, help="HBM bytes/launch measured elsewhere (overrides --traffic)")
    ap.add_argument("--traffic", choices=["live", "static"], default="live",
                    help="roofline.traffic at N=1: live = two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same frame in child "
                         "processes after the timed region (about 15 s; falls back to static if rocprofv3 is missing or fails); "
                         "static = the committed profiles/*_pmc_summary.json")
    args = ap.parse_args()
    if args.warmup_ms is None:
        given = any(x == "--warmup" or x.startswith("--warmup=") for x in sys.argv[1:])
        args.warmup_ms = 0.0 if given else 40.0
    return args


def load_workload(args, bs):
    cfg_obj = workload_config(bs, "default-aa" if args.workload == "animation" else args.workload)
    cfg = cfg_obj.to_bs_config()
    frames_cfg = frames_obj = None
    if args.workload == "animation":
        anim = bs.Animation.from_file(os.path.join(ROOT, "animations", "default-ani.yaml"))
        anim.nFrames = 600  # BASELINE configs[4] (the file itself says 375)
        bs.validate_keyframes(anim.keyframes)
        frames_obj = bs.generate_frames(anim)
        frames_cfg = [c.to_bs_config() for c in frames_obj]
        cfg = frames_cfg[0]
    return cfg_obj, cfg, frames_cfg, frames_obj


def run_ranks(args):
    """One process per GPU (this process is one rank; torch.distributed.run or the driver's launcher set the env)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import blackstar_amd as bs
    from blackstar_amd import _lib, synthetic

    sw = Stopwatch()
    sw.seconds["imports"] = round(time.perf_counter() - T_START, 3)
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
    devices_or_die(world, ndev, os.environ.get("BLACKSTAR_BENCH_ALLOW_OVERSUBSCRIBE") == "1")
    if backend != "nccl":
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    # one process per GPU: this rank's threads on the GPU's NUMA node (BLACKSTAR_NUMA_BIND=0 leaves them alone, like the library's own binding)
    host_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None   # (what the CPU baseline may use: cpu_baseline)
    binding = bind_rank_to_gpu_node(torch, local_rank) if os.environ.get("BLACKSTAR_NUMA_BIND", "1") != "0" else {"bound": False, "why": "BLACKSTAR_NUMA_BIND=0"}
    rccl = None
    # BLACKSTAR_BENCH_FORCE_DIST=1 (set by `--launcher torchrun --gpus 1`): one rank still goes through init_process_group, the all_gathers,
    # the barrier, all_gather_object, --gather's dist.gather and destroy_process_group -- the RCCL branch executed on ONE GPU, so that an
    # API misuse surfaces on a 1-GPU box instead of on the 8-GPU lease.  The numbers are those of world 1.
    dist_on = world > 1 or os.environ.get("BLACKSTAR_BENCH_FORCE_DIST") == "1"
    if dist_on:
        if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:   # no launcher around a forced world-1 run
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # Untimed: one all_gather of (rank, PCI bus of the rank's device) -- the path itself has no collective, so this is what shows
        # that RCCL saw N ranks on N distinct devices (and it establishes the communicator outside the timed region).
        dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        bus = pci_bus_of(torch, local_rank)
        me = torch.tensor([rank, -1 if bus is None else bus], dtype=torch.int64, device=dev)
        got = [torch.empty_like(me) for _ in range(world)]
        dist.all_gather(got, me)
        torch.cuda.synchronize()
        rccl = {"backend": "RCCL (torch.distributed nccl)" if backend == "nccl" else backend, "ranks": sorted(int(g[0]) for g in got),
                "pci_bus_per_rank": [f"{int(g[1]):02x}" if int(g[1]) >= 0 else None for g in got],
                "distinct_devices": len({int(g[1]) for g in got}),
                "version": ".".join(map(str, torch.cuda.nccl.version())) if backend == "nccl" else None}

    t_setup = time.perf_counter()
    cfg_obj, cfg, frames_cfg, frames_obj = load_workload(args, bs)
    if args.form == "split":  # the split form has its own frame (configs[3]); the warm-up launches use it too
        cfg_obj = workload_config(bs, "lensing-4k")
        cfg = cfg_obj.to_bs_config()
    W, H = cfg["width"], cfg["height"]
    with_stars = WORKLOADS[args.workload]["stars"] or args.form == "split"   # configs[1] is "no starmap": an empty star set
    star_bytes = synthetic.catalogue_bytes(args.catalogue) if with_stars else None
    stars = bs.read_map(star_bytes) if with_stars else bs.read_map(bytes(28))   # (28 header bytes, no records)
    if world > ndev:  # smoke mode: ranks share a device, and every context would set the SAME few CUs aside for its post stage
        os.environ.setdefault("BLACKSTAR_POST_CUS", "0")
    tree = bs.StarTree(stars, device=local_rank)
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
        # every rank's GPU work has finished BEFORE any rank passes the barrier (a rank that is still executing must not overlap a
        # neighbour's timed region when ranks share a device in the smoke modes), and RCCL's own barrier kernel has finished after it
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    def all_ranks(x):  # every rank's value of a float, in rank order
        if not dist_on:
            return [float(x)]
        dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        return [float(v.item()) for v in allt]

    def gather_objs(o):  # every rank's (small, picklable) object, in rank order -- validation only, never inside a timed region
        if not dist_on:
            return [o]
        objs = [None] * world
        dist.all_gather_object(objs, o)
        return objs

    def gather_to_root():  # optional (--gather): each rank's finished frame goes to rank 0 over xGMI
        src = out if backend == "nccl" else out.cpu()
        gathered = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
        dist.gather(src, gathered, dst=0)
        return gathered

    def my_frames(n):  # the Config objects of this rank's next n frames (the d2h forms take Configs: bloom parameters are part of a frame)
        if frames_obj is None:
            return [cfg_obj] * n
        return [frames_obj[(j * world + rank) % len(frames_obj)] for j in range(n)]

    resident = args.form in ("all", "resident")
    dt_local = kernel_ms = None
    sw.seconds["catalogue_and_context"] = round(time.perf_counter() - t_setup, 3)
    t_timed = time.perf_counter()
    if resident:
        # Everything the host has to do around the timed region is done BEFORE the warm-up steps: the chip drops its clocks within a
        # millisecond of idling and takes ~8 launches (35 ms) to bring them back (kernel_ms_each: 5.5, 5.1, 4.9, 4.7, 4.6, 4.5 ... 4.3 ms
        # after a 40 ms collector pause between the warm-up and the timed steps), so nothing but the fence stands between the two.
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        gc.collect()   # no collector pause inside the 90 ms of the timed region (the catalogue's temporaries are garbage by now)
        gc.disable()
        t_w = time.perf_counter()
        for _ in range(args.warmup):
            step()
        if args.warmup_ms > 0 and args.warmup > 0:   # (default flags only: see --warmup-ms)
            for _ in range(4):   # (the first launches run on cold clocks and over-state a step: look again after each batch)
                torch.cuda.synchronize()
                spent = (time.perf_counter() - t_w) * 1e3
                if spent >= args.warmup_ms:
                    break
                more = min(2000, int((args.warmup_ms - spent) / (spent / args.warmup)) + 1)
                for _ in range(more):
                    step()
                args.warmup += more   # (the line reports the warm-up steps that were really run)
        if dist_on and args.gather:
            gather_to_root()  # also establishes RCCL's point-to-point channels outside the timed region
        fence()
        t0 = time.perf_counter()
        for a, b in ev:
            s = lanes[counter["k"] % n_streams][1]
            a.record(s)
            step()
            b.record(s)
        t_gather = None
        if dist_on and args.gather:
            torch.cuda.synchronize()
            tg = time.perf_counter()
            gather_to_root()
            torch.cuda.synchronize()
            t_gather = time.perf_counter() - tg
        fence()
        dt_local = time.perf_counter() - t0
        gc.enable()
        st = tree.stats()
        kernel_each = [float(a.elapsed_time(b)) for a, b in ev]  # per launch incl. the 64-B counter memset/copy nodes
        kernel_ms = float(np.mean(kernel_each))
        allt = all_ranks(dt_local)
        per_rank_ms = [t / args.steps * 1e3 for t in allt]
        dt = max(allt)  # MAX over ranks
    else:  # --form batch | rgb8-batch | png-batch | png-files | split: that form IS the timed region (warm-up inside d2h_forms, same fence discipline)
        for _ in range(max(1, args.warmup)):
            step()
        fence()
        st = tree.stats()
        t_gather = None

    sw.seconds["warmup_and_timed_region"] = round(time.perf_counter() - t_timed, 3)
    # BASELINE configs[1] and configs[3] beside the headline, while the clocks are still up (N = 1, the default workload)
    per_config = None
    if world == 1 and resident and args.workload == "default-aa" and not args.no_boundary:
        with sw.leg("per_config"):
            per_config = optional_leg("per_config", True, lambda: per_config_block(bs, torch, np, _lib, tree, args, local_rank))

    # Untimed validation: every rank renders the SAME frame once more (the animation: its frame 0); the frames must be bit-identical
    validation = None
    if not args.no_validate:
        def validate():
            vcfg = cfg if frames_cfg is None else frames_cfg[0]
            digests = []
            try:   # a rank that fails here still takes part in the gather below (with a digest no other rank can have): the line says invalid, nobody hangs
                for _ in range(2 if rank == 0 else 1):   # rank 0 twice: run-to-run determinism
                    out.zero_()
                    bs.render_device(vcfg, tree, out.data_ptr(), out.numel(), stream.cuda_stream)
                    torch.cuda.synchronize()
                    digests.append(frame_digest(np, out))
                vst = tree.stats()
                mine = (digests[0], {k: int(vst[k]) for k in COUNTERS})
            except Exception as e:
                print(f"bench.py: validation render failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                mine = (f"rank {rank} failed: {type(e).__name__}: {e}", {k: 0 for k in COUNTERS})
            got = gather_objs(mine)
            what = "the workload's frame" if frames_cfg is None else "frame 0 of the animation"
            return validation_block(got, what + ", rendered once more on every rank after the timed region (untimed)", digests[1] if rank == 0 and len(digests) > 1 else None)
        with sw.leg("validation"):
            validation = optional_leg("validation", world == 1, validate)

    d2h = None
    want = {"all": ["batch", "rgb8-batch", "png-batch", "png-files"] + (["split"] if with_stars else []), "resident": []}.get(args.form, [args.form])
    if want:
        with sw.leg("with_d2h"):
            d2h = optional_leg("with_d2h", world == 1 and resident,
                               lambda: d2h_forms(bs, np, [tree], my_frames(args.steps), W, H, world, want, fence, lambda x: max(all_ranks(x)),
                                                 same_frames=frames_obj is None, all_ranks=all_ranks if dist_on else None,
                                                 split=lambda: split_leg(bs, np, [tree], rank, world, fence, lambda x: max(all_ranks(x)), gather_objs),
                                                 extras=rank == 0))   # (host probes on the rank that prints: eight ranks probing one file system at once would measure each other)

    def sustained_block():
        n_sus = args.sustained_frames // 50 * 50
        fence()
        failed = None   # (a rank that fails here still goes through the fence and the collectives below: nobody hangs, the leg reports the error)
        per_dev, devices = [{"ms_per_frame": float("inf")}], [None]
        try:
            with DeviceSampler([pci_bus_of(torch, local_rank)]) as smp:
                wall, per_dev = sustained_leg(bs, torch, np, [tree], [cfg if frames_cfg is None else frames_cfg[rank % len(frames_cfg)]], [out], [stream],
                                              [local_rank], n_sus)
            devices = smp.summary()
        except Exception as e:
            failed = f"{type(e).__name__}: {e}"
            print(f"bench.py: sustained leg failed on rank {rank}: {failed}", file=sys.stderr, flush=True)
        fence()
        ms_all = all_ranks(per_dev[0]["ms_per_frame"])
        if dist_on:  # every rank sampled its own device: collect them in rank order
            objs = [None] * world
            dist.all_gather_object(objs, devices)
            devices = [d for o in objs for d in (o or [None])]
        if failed is not None or max(ms_all) == float("inf"):
            return {"error": failed or "another rank failed"}
        return dict(per_dev[0], frames=n_sus, Mpixel_s=world * W * H / max(ms_all) / 1e3, per_rank_ms_per_frame=ms_all,
                    device=devices, note="back-to-back launches of the same frame on one stream per GPU, all ranks at once; "
                                         "Mpixel_s from the slowest rank; device = sampled sclk / power, one entry per rank")

    sustained = None
    if args.sustained_frames >= 50 and resident:
        with sw.leg("sustained"):
            sustained = optional_leg("sustained", world == 1, sustained_block)

    if rank == 0:
        frames = args.steps * world
        peak = None
        if world == 1 and not args.no_boundary and frames_cfg is None and resident:
            peak = optional_leg("measure_peak", True, lambda: measure_peak(tree, _lib))
            peak = None if isinstance(peak, dict) and "error" in peak else peak
        extra = {"backend": ("RCCL (nccl)" if backend == "nccl" else backend) + ("" if world > 1 else " -- forced at world 1: BLACKSTAR_BENCH_FORCE_DIST")
                            if dist_on else "none (single rank)",
                 "devices_visible": ndev, "oversubscribed": world > ndev, "launches_in_flight_per_gpu": n_streams,
                 "catalogue": args.catalogue, "n_stars": int(len(stars)), "effective_mode": ["strict", "fast"][int(st["effective_mode"])],
                 "rank0_cpu_binding": binding}
        if n_streams > 1:
            extra["launches_in_flight_note"] = ("consecutive frames alternate between two streams and share the GPU, so kernel_ms "
                                                "(per-launch event time) exceeds ms_per_step")
        launcher = "torchrun-env (one process per GPU)" if world > 1 or "WORLD_SIZE" in os.environ else "single-process"
        if resident:
            value = frames * W * H / dt / 1e6
        else:  # the named d2h form is the result
            key = args.form.replace("-", "_")
            if "Mpixel_s" not in d2h[key]:
                raise SystemExit(f"bench.py --form {args.form}: {d2h[key]}")
            value, dt = d2h[key]["Mpixel_s"], d2h[key]["seconds"]
            kernel_ms = dt / args.steps * 1e3
            extra["image"] = d2h[key]["note"]
            per_rank_ms = [kernel_ms] * world
        res = result_line(args, world, launcher, value, dt, W, H, frames_cfg, st, kernel_ms, extra, peak, catalogue_note(args, len(stars)))
        if args.form == "split":
            split_headline(args, res, d2h["split"], world)
        res["per_rank_ms_per_step"] = per_rank_ms
        label_roofline_scope(res, world, per_rank_ms)
        if legs_failed(d2h):
            res["legs_failed"] = legs_failed(d2h)
        if validation is not None:
            res["validation"] = validation
            res["valid"] = bool(validation.get("valid", False)) and forms_valid(d2h) and not legs_failed(d2h)
        if per_config is not None:
            res["per_config"] = per_config
        if resident and n_streams == 1:
            res["kernel_ms_each"] = [round(x, 4) for x in kernel_each]  # the timed launches one by one (rank 0): a clock ramp after the idle start-up shows here
        if rccl is not None:
            res["rccl"] = rccl
        if t_gather is not None:
            res["gather_ms"] = t_gather * 1e3
            res["config"]["gather"] = "dist.gather of every rank's last frame to rank 0 inside the timed region"
        if d2h:
            res["with_d2h"] = d2h
        if sustained:
            res["sustained"] = sustained
        if world == 1 and not args.no_boundary and frames_cfg is None and resident:
            with sw.leg("boundary_and_strict"):
                both = optional_leg("boundary", True, lambda: boundary_numbers(bs, _lib, tree, cfg_obj, cfg, args, torch, out, stream))
            res["boundary"], res["strict"] = both if isinstance(both, tuple) else (both, both)
        if world == 1 and args.cpu_seconds > 0:
            def gpu_render(c, with_stars_, mode):   # the product, through bs_render (C ABI), in the named arithmetic; the oracle only checks it
                t = tree if with_stars_ else bs.StarTree(None, device=local_rank)
                before = t.get_mode()
                try:
                    t.set_mode(_lib.BS_MODE_FAST if mode == "fast" else _lib.BS_MODE_STRICT)
                    image = bs.render(c, t)
                    return image, t.stats()
                finally:
                    if t is tree:
                        t.set_mode(before)
                    else:
                        t.close()
            with sw.leg("cpu_baseline_and_parity"):
                res["cpu_baseline"] = optional_leg("cpu_baseline", True, lambda: cpu_baseline(cfg, star_bytes, args.cpu_seconds, gpu_render, np, WORKLOADS[args.workload]["baseline"], host_cpus=host_cpus))
            if isinstance(res["cpu_baseline"], dict) and "parity_ok" in res["cpu_baseline"] and "valid" in res:
                res["valid"] = bool(res["valid"]) and bool(res["cpu_baseline"]["parity_ok"])   # a fast frame that differs from the reference's is not a result
        if world == 1 and resident and args.traffic == "live" and args.traffic_bytes is None and not args.no_boundary and args.workload == "default-aa":
            # LAST: the profiler's child processes run after every timed leg of this process (a PMC session may leave the device in
            # another clock state for a while), and only the counter values are taken from them
            with sw.leg("pmc_passes"):
                try:
                    args.traffic_live = pmc_traffic_live(args.mode, args.catalogue)
                except Exception as e:  # nothing about the profiler may cost the line that has already been measured
                    args.traffic_live = (None, f"{type(e).__name__}: {e}")
            fresh = roofline_block(args, st, kernel_ms, W, H)
            res["roofline"].update({k: fresh[k] for k in ("traffic", "traffic_kind", "traffic_source", "frac_cycles", "busy_cycles_per_launch",
                                                          "flop_per_cycle_peak", "frac_cycles_detail", "sclk_MHz_implied") if k in fresh})
        sw.seconds["total_so_far"] = round(time.perf_counter() - T_START, 3)
        res["leg_seconds"] = sw.seconds   # where this process's wall time went (the timed region itself is ms_per_step x steps)
        emit(res)
    tree.close()
    if dist_on:
        dist.destroy_process_group()


def run_single_process(args):
    """N GPUs from ONE process: one bs_ctx + one output image + one stream per device, every step enqueues one frame on each
    (bs_render_device is asynchronous, so one host thread keeps N GPUs busy); no collective of any kind.  The with_d2h forms are ONE
    call of bs_render_batch / bs_render_rgb8_batch over all N contexts: the product's own multi-GPU API (one host thread per context)."""
    import numpy as np
    import torch

    import blackstar_amd as bs
    from blackstar_amd import _lib, synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    world = args.gpus
    ndev = torch.cuda.device_count()
    devices_or_die(world, ndev, os.environ.get("BLACKSTAR_BENCH_ALLOW_OVERSUBSCRIBE") == "1")
    devs = [i % ndev for i in range(world)]
    cfg_obj, cfg, frames_cfg, frames_obj = load_workload(args, bs)
    if args.form == "split":  # the split form has its own frame (configs[3]); the warm-up launches use it too
        cfg_obj = workload_config(bs, "lensing-4k")
        cfg = cfg_obj.to_bs_config()
    W, H = cfg["width"], cfg["height"]
    with_stars = WORKLOADS[args.workload]["stars"] or args.form == "split"   # configs[1] is "no starmap": an empty star set
    stars = bs.read_map(synthetic.catalogue_bytes(args.catalogue)) if with_stars else bs.read_map(bytes(28))
    n_streams = args.streams or (2 if frames_cfg is not None else 1)
    trees, outs, streams, lanes = [], [], [], []
    if world > ndev:  # smoke mode: contexts share a device, and every one of them would set the SAME few CUs aside for its post stage
        os.environ.setdefault("BLACKSTAR_POST_CUS", "0")
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

    resident = args.form in ("all", "resident")
    ev = []
    if resident:  # (host work first, then warm-up, fence, timed steps: see run_ranks)
        for k in range(world):
            with torch.cuda.device(devs[k]):
                ev.append([(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)])
        gc.collect()
        gc.disable()
    for _ in range(args.warmup if resident else max(1, args.warmup)):
        for k in range(world):
            step(k)
        counter["i"] += 1
    if args.gather:
        gather_to_root()
    fence()
    t_gather = None
    if resident:
        t0 = time.perf_counter()
        for s in range(args.steps):
            for k in range(world):
                with torch.cuda.device(devs[k]):
                    st_ = lane(k)[1]
                    ev[k][s][0].record(st_)
                    step(k)
                    ev[k][s][1].record(st_)
            counter["i"] += 1
        if args.gather:
            fence()
            tg = time.perf_counter()
            gather_to_root()
            fence()
            t_gather = time.perf_counter() - tg
        fence()
        dt = time.perf_counter() - t0  # one clock for all devices: this IS the max over "ranks"
        gc.enable()
        kms = [[a.elapsed_time(b) for a, b in ev[k]] for k in range(world)]
        per_rank_ms = [dt / args.steps * 1e3] * world if n_streams > 1 else [ev[k][0][0].elapsed_time(ev[k][-1][1]) / args.steps for k in range(world)]
        kernel_ms = float(np.mean(kms[0]))
        value = args.steps * world * W * H / dt / 1e6
    st = trees[0].stats()

    # Untimed validation: every device renders the SAME frame once more (the animation: its frame 0); the frames must be bit-identical
    validation = None
    if not args.no_validate:
        def validate():
            vcfg = cfg if frames_cfg is None else frames_cfg[0]
            got, repeat = [], None
            for k in range(world):
                for rep in range(2 if k == 0 else 1):   # device 0 twice: run-to-run determinism
                    with torch.cuda.device(devs[k]), torch.cuda.stream(streams[k]):   # (the clear on the stream the render is enqueued on)
                        outs[k].zero_()
                    bs.render_device(vcfg, trees[k], outs[k].data_ptr(), outs[k].numel(), streams[k].cuda_stream)
                    torch.cuda.synchronize(devs[k])
                    d = frame_digest(np, outs[k])
                    if rep == 0:
                        got.append((d, {c: int(v) for c, v in trees[k].stats().items() if c in COUNTERS}))
                    else:
                        repeat = d
            what = "the workload's frame" if frames_cfg is None else "frame 0 of the animation"
            return validation_block(got, what + ", rendered once more on every context after the timed region (untimed)", repeat)
        validation = optional_leg("validation", True, validate)

    d2h = None
    want = {"all": ["batch", "rgb8-batch", "png-batch", "png-files"] + (["split"] if with_stars else []), "resident": []}.get(args.form, [args.form])
    if want:  # frame i on context i % world, args.steps frames per context, ONE call over all contexts
        n = args.steps * world
        objs = [cfg_obj] * n if frames_obj is None else [frames_obj[i % len(frames_obj)] for i in range(n)]
        d2h = optional_leg("with_d2h", resident, lambda: d2h_forms(bs, np, trees, objs, W, H, 1, want, fence, lambda x: x, same_frames=frames_obj is None,
                                                                   split=lambda: split_leg(bs, np, trees, 0, 1, fence, lambda x: x, lambda o: [o])))

    # (the sustained leg with its clock sampler belongs to the one-process-per-GPU form: run_ranks)
    extra = {"backend": "none (one process, one bs_ctx + stream per device; frames never leave their GPU)",
             "devices_visible": ndev, "oversubscribed": world > ndev, "devices": devs, "launches_in_flight_per_gpu": n_streams,
             "catalogue": args.catalogue, "n_stars": int(len(stars)), "effective_mode": ["strict", "fast"][int(st["effective_mode"])]}
    if not resident:
        key = args.form.replace("-", "_")
        if "Mpixel_s" not in d2h[key]:
            raise SystemExit(f"bench.py --form {args.form}: {d2h[key]}")
        value, dt = d2h[key]["Mpixel_s"], d2h[key]["seconds"]
        kernel_ms = dt / args.steps * 1e3
        per_rank_ms, kms = [kernel_ms] * world, None
        extra["image"] = d2h[key]["note"]
    res = result_line(args, world, "single-process (N contexts)", value, dt, W, H, frames_cfg, st, kernel_ms, extra, None, catalogue_note(args, len(stars)))
    if args.form == "split":
        split_headline(args, res, d2h["split"], world)
    res["per_rank_ms_per_step"] = per_rank_ms
    label_roofline_scope(res, world, per_rank_ms)
    if legs_failed(d2h):
        res["legs_failed"] = legs_failed(d2h)
    if validation is not None:
        res["validation"] = validation
        res["valid"] = bool(validation.get("valid", False)) and forms_valid(d2h) and not legs_failed(d2h)
    if kms:
        res["per_rank_kernel_ms"] = [float(np.mean(x)) for x in kms]
    if t_gather is not None:
        res["gather_ms"] = t_gather * 1e3
        res["config"]["gather"] = "peer copy of every device's last frame to device 0 inside the timed region"
    if d2h:
        res["with_d2h"] = d2h
    emit(res)
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
    if args.gpus == 1:
        env["BLACKSTAR_BENCH_FORCE_DIST"] = "1"   # one rank, and still every torch.distributed call of the N > 1 path (see run_ranks)
    if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
        env.setdefault("BLACKSTAR_BENCH_BACKEND", "gloo")  # smoke mode: ranks share devices, RCCL needs one device per rank
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.launcher != "torchrun" or "WORLD_SIZE" in os.environ:   # (the re-exec form's parent prints nothing itself: its ranks claim theirs)
        claim_stdout()
    if "WORLD_SIZE" in os.environ:
        run_ranks(args)
    elif args.launcher == "torchrun":
        reexec_under_torchrun(args)
    elif args.gpus == 1:
        run_ranks(args)
    else:
        run_single_process(args)


if __name__ == "__main__":
    main()
