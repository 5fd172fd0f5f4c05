"""bench_legs.py -- what bench.py measures BESIDE its headline value, and how the result line is put together: the workloads table, the
delivered forms (with_d2h), the split leg, per_config, validation, sustained, boundary, roofline / PMC traffic, the device sampler
(the cpu_baseline leg -- the only place the oracle is touched -- stays in bench.py).  bench.py (the driver's contract: CLI, the timed region, one JSON line) imports everything from here; tests reach the same names
through `bench.`.  Nothing here is product code."""
import json
import os
import subprocess
import sys
import time

__all__ = ["ROOT", "FLOP_PER_STEP", "PEAK_FP64_VALU_TFLOPS", "PEAK_HBM_GBS", "LOOP_VALU", "WORKLOAD_C3", "WORKLOAD_C5", "WORKLOADS", "workload_config", "CATALOGUES", "catalogue_note", "DeviceSampler", "pci_bus_of", "bind_rank_to_gpu_node", "pmc_traffic", "pmc_traffic_live", "cycles_view", "d2h_forms", "png_files_leg", "write_rate_probe", "host_block", "predict_frames_8_gpus", "Stopwatch", "sustained_leg", "frame_digest", "digest_as_float", "WARM_PER_CONTEXT", "TIMED_CALLS", "pcie_zero_copy_probe", "COUNTERS", "validation_block", "per_config_block", "predict_bands", "split_leg", "optional_leg", "roofline_block", "measure_peak", "boundary_numbers", "forms_valid", "label_roofline_scope", "devices_or_die", "legs_failed", "split_headline", "result_line"]

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PER_CONFIG_WARM_MS = 45.0   # untimed launches before the timed ones of per_config_block (the clocks need ~35 ms of work to come back from idle)
FLOP_PER_STEP = 145  # 130 (rk4, src/Raytracer.hs:113-134) + 15 (findColor where-bindings, :100-102)
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X FP64 vector, FMA = 2 flop (= 1/2 of the guide's 157.3 TF FP32 vector peak)
PEAK_HBM_GBS = 8000.0
# VALU instructions the stepping loop issues per RK4 step of a wavefront (ISA count, scripts/isa_hot_blocks.py; static):
# full-rate f64 ops and quarter-rate transcendental seeds (v_rsq_f64 / v_rcp_f64 occupy the pipe for 4 issue slots).  FAST: the in-line path
# of csrc/fast_loop_asm.h (the r^-5 of stages 1 and 3 from a neighbouring evaluation by a series); the few steps per hundred that visit an
# out-of-line block (7 more + 1 seed each) are not counted, so valu_issue_frac UNDER-states the issue rate by about 1 %.
LOOP_VALU = {"fast": {"full_rate": 65, "quarter_rate": 2}, "strict": {"full_rate": 178, "quarter_rate": 8}}
WORKLOAD_C3 = ("scenes/default-aa.yaml 1920x1080, 4x supersample (8,294,400 rays/frame), {cat}, "
               "direction-grid star lookup (BASELINE configs[2])")
WORKLOAD_C5 = ("animations/default-ani.yaml, nFrames=600, 1920x1080, 4x supersample, {cat}, "
               "frame i on rank i % N (BASELINE configs[4]); roofline figures refer to the LAST frame rendered")
# The single-frame BASELINE configs that fit one GPU: scene file, resolution override, whether the star map is part of the config.
WORKLOADS = {
    "default-aa": {"scene": "default-aa.yaml", "resolution": None, "stars": True, "baseline": "configs[2]", "label": WORKLOAD_C3,
                   "metric": "Mpixel/s (geodesic rays/s) on default-aa.yaml"},
    "default": {"scene": "default.yaml", "resolution": None, "stars": False, "baseline": "configs[1]",
                "label": "scenes/default.yaml 1920x1080, no supersampling (2,073,600 rays/frame), NO star map: disk + horizon only (BASELINE configs[1])",
                "metric": "Mpixel/s (geodesic rays/s) on default.yaml, no star map"},
    "lensing-4k": {"scene": "lensing-disk.yaml", "resolution": (3840, 2160), "stars": True, "baseline": "configs[3]",
                   "label": "scenes/lensing-disk.yaml at 3840x2160, 4x supersample (33,177,600 rays/frame), {cat}, direction-grid star lookup "
                            "(BASELINE configs[3]: close-orbit long geodesics)",
                   "metric": "Mpixel/s (geodesic rays/s) on lensing-disk.yaml at 3840x2160"},
    "animation": {"label": WORKLOAD_C5, "baseline": "configs[4]", "stars": True, "metric": "Mpixel/s (geodesic rays/s) on default-aa.yaml"},
}


def workload_config(bs, name):
    """The Config of a single-frame workload: the scene file as the reference ships it, with BASELINE's resolution override."""
    w = WORKLOADS[name]
    cfg = bs.Config.from_file(os.path.join(ROOT, "scenes", w["scene"]))
    return cfg.with_resolution(*w["resolution"]) if w["resolution"] else cfg
CATALOGUES = {"synthetic": "470k-star synthetic PPM-layout catalogue (uniform sky, SURVEY 8d recipe)",
              "clustered": "686k-star NON-uniform synthetic PPM-layout catalogue (the 470k uniform stars + 3000 clusters of 6..40 stars "
                           "inside 0.001 rad + a band at 10x the mean density; blackstar_amd/synthetic.py)"}


def catalogue_note(args, n_stars):
    return CATALOGUES.get(args.catalogue, f"REAL catalogue file {os.path.basename(args.catalogue)} ({n_stars} stars; not the BASELINE input: reported separately)")


class DeviceSampler:
    """Mean shader clock and package power of the given devices while a leg runs: amdgpu's hwmon files (freq1_input = sclk in Hz,
    power1_input / power1_average = package power in uW), matched to HIP devices by PCI bus number, read by one thread at ~25 Hz.
    The box's sysfs lists every GPU of the host, other tenants' included: a device whose bus cannot be matched is not reported."""

    def __init__(self, pci_bus_ids, sysfs="/sys/class/drm"):
        import glob
        found = {}
        for hw in glob.glob(os.path.join(sysfs, "card[0-9]*/device/hwmon/hwmon*")):
            try:
                bus = int(os.path.basename(os.path.realpath(os.path.join(hw, "..", ".."))).split(":")[1], 16)
            except (IndexError, ValueError):
                continue
            files = {}
            if os.path.exists(os.path.join(hw, "freq1_input")):
                files["sclk_Hz"] = os.path.join(hw, "freq1_input")
            for f in ("power1_input", "power1_average"):
                if os.path.exists(os.path.join(hw, f)):
                    files["power_uW"] = os.path.join(hw, f)
                    break
            if files:
                found[bus] = files
        self.cards = [(b, found[b]) for b in dict.fromkeys(pci_bus_ids) if b in found]
        self.samples = [{k: [] for k in files} for _, files in self.cards]
        self._stop = self._t = None

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                for (_, files), acc in zip(self.cards, self.samples):
                    for k, fn in files.items():
                        try:
                            with open(fn) as f:
                                acc[k].append(float(f.read().split()[0]))
                        except (OSError, ValueError, IndexError):
                            pass
                self._stop.wait(0.04)
        if self.cards:
            self._t = threading.Thread(target=loop, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._t:
            self._t.join()

    def summary(self):
        out = []
        for (bus, _), acc in zip(self.cards, self.samples):
            d = {"pci_bus": f"{bus:02x}", "samples": max((len(v) for v in acc.values()), default=0)}
            if acc.get("sclk_Hz"):
                d["sclk_MHz_mean"] = sum(acc["sclk_Hz"]) / len(acc["sclk_Hz"]) / 1e6
                d["sclk_MHz_min"] = min(acc["sclk_Hz"]) / 1e6
            if acc.get("power_uW"):
                d["power_W_mean"] = sum(acc["power_uW"]) / len(acc["power_uW"]) / 1e6
                d["power_W_max"] = max(acc["power_uW"]) / 1e6
            out.append(d)
        return out or None


def bind_rank_to_gpu_node(torch, d):
    """One process per GPU: run this rank's threads on the CPUs of the NUMA node its GPU hangs off (sysfs local_cpulist of the device's PCI
    function, restricted to what the process may use) -- on a two-socket 8-GPU node half of the ranks would otherwise drive a GPU across
    the socket link.  The library binds the threads IT starts itself (bs::NumaBind); this covers the rank's own Python thread, which
    enqueues the resident form's launches.  Returns what was done, for the result line; never raises."""
    try:
        pr = torch.cuda.get_device_properties(d)
        bdf = f"{int(pr.pci_domain_id):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(base + "/numa_node").read().split()[0])
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        use = cpus & allowed
        if node < 0 or not use or use == allowed:
            return {"pci": bdf, "numa_node": node, "bound": False, "why": "the host gives no node for the device, or the node is all this process may use"}
        os.sched_setaffinity(0, use)
        return {"pci": bdf, "numa_node": node, "bound": True, "cpus": len(use)}
    except (OSError, ValueError, AttributeError, RuntimeError) as e:
        return {"bound": False, "why": f"{type(e).__name__}: {e}"}


def pci_bus_of(torch, d):
    try:
        return int(torch.cuda.get_device_properties(d).pci_bus_id)
    except (AttributeError, RuntimeError, ValueError):
        return None


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


N_XCD, N_CU, SIMD_PER_CU, LANES_PER_SIMD_F64 = 8, 256, 4, 16   # MI355X: 8 XCDs x 32 CUs, 4 SIMDs per CU, 16 f64 lanes per SIMD and cycle


def pmc_traffic_live(mode, catalogue, timeout_s=60):
    """HBM bytes per launch of the trace kernel MEASURED NOW: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE -- separate
    runs, as MI355X_MICROARCH.md's HBM section prescribes: the two do not fit the TCC's slots together) over scripts/prof_frame.py, which
    renders the same frame three times through the C ABI in a child process.  The WRITE_SIZE pass also carries GRBM_GUI_ACTIVE (the GRBM
    block has slots of its own): the launch's busy cycles, summed over the 8 XCDs -- the clock-independent view of the same launch
    (roofline.frac_cycles).  Counters cannot be read from inside an un-profiled process, hence the children; timing is never taken from
    them.  Returns (bytes, detail) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    got = {}
    work = tempfile.mkdtemp(prefix="bs_pmc_", dir="/tmp")
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE", "GRBM_GUI_ACTIVE")):
            out = os.path.join(work, counters[0])
            cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "-o", "t", "--",
                   sys.executable, os.path.join(ROOT, "scripts", "prof_frame.py"), "--mode", mode, "--stars", catalogue, "--frames", "3"]
            try:
                r = subprocess.run(cmd, cwd=work, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, f"rocprofv3 --pmc {counters[0]}: {type(e).__name__}"
            vals = {c: [] for c in counters}
            for fn in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(fn) as f:
                    for row in csv.DictReader(f):
                        if row.get("Counter_Name") in vals and "trace_frame" in row.get("Kernel_Name", ""):
                            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals[counters[0]]:
                return None, f"rocprofv3 --pmc {counters[0]}: rc {r.returncode}, {len(vals[counters[0]])} samples"
            for c, v in vals.items():
                if v:
                    got[c] = sum(v) / len(v)
            if "GRBM_GUI_ACTIVE" in counters:
                ns = []
                for fn in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                    with open(fn) as f:
                        ns += [int(row["End_Timestamp"]) - int(row["Start_Timestamp"]) for row in csv.DictReader(f) if "trace_frame" in row.get("Kernel_Name", "")]
                if ns:
                    got["kernel_ns_in_pass"] = sum(ns) / len(ns)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # KiB units; FETCH_SIZE doubled: the guide's gfx950 correction (an upper bound for this kernel's 32-byte star-grid reads)
    detail = {"FETCH_SIZE_KiB": got["FETCH_SIZE"], "WRITE_SIZE_KiB": got["WRITE_SIZE"], "launches_per_pass": 3}
    if "GRBM_GUI_ACTIVE" in got:
        detail["GRBM_GUI_ACTIVE"] = got["GRBM_GUI_ACTIVE"]
        if got.get("kernel_ns_in_pass"):
            detail["kernel_ms_in_profiled_pass"] = got["kernel_ns_in_pass"] / 1e6
            detail["sclk_MHz_in_profiled_pass"] = got["GRBM_GUI_ACTIVE"] / N_XCD / got["kernel_ns_in_pass"] * 1e3
    return (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0, detail


def cycles_view(flop_per_launch, gui_active, kernel_ms=None):
    """The launch against the FP64 vector roof in CYCLES instead of seconds: flop / (busy cycles x 256 CUs x 4 SIMDs x 16 lanes x 2 flop
    per FMA).  GRBM_GUI_ACTIVE is summed over the 8 XCDs, which all stay busy for the whole launch, so a launch's cycles = the counter / 8.
    The same kernel on a box whose power cap holds 2.27 GHz and on one that holds 2.36 gives the same frac_cycles; `frac` (seconds) differs
    by the clock ratio.  kernel_ms (the un-profiled launch time of this run) turns the cycles into the clock the chip averaged: the peak
    78.6 TFLOP/s is this roof at 2.4 GHz."""
    cycles = gui_active / N_XCD
    per_cycle = N_CU * SIMD_PER_CU * LANES_PER_SIMD_F64 * 2
    out = {"frac_cycles": flop_per_launch / (cycles * per_cycle), "busy_cycles_per_launch": cycles, "flop_per_cycle_peak": per_cycle,
           "frac_cycles_detail": "145 flop x executed RK4 steps / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CU x 4 SIMD x 16 lanes x 2): clock-independent"}
    if kernel_ms:
        out["sclk_MHz_implied"] = cycles / (kernel_ms * 1e-3) / 1e6   # cycles of the profiled launch over the un-profiled duration
    return out


class Stopwatch:
    """Wall seconds per leg of a bench run (`leg_seconds` in the result line): where the driver's run time goes."""

    def __init__(self):
        self.seconds = {}

    def leg(self, name):
        sw = self

        class _Leg:
            def __enter__(self):
                self.t0 = time.perf_counter()

            def __exit__(self, *exc):
                sw.seconds[name] = round(sw.seconds.get(name, 0.0) + time.perf_counter() - self.t0, 3)
        return _Leg()


def write_rate_probe(directory, nbytes, threads=(1, 8), files_per_thread=24):
    """What the HOST can do for bs_render_png_files' writers, measured in the target directory: `threads` concurrent threads each create /
    write / close files_per_thread files of nbytes bytes (a frame's PNG), like the library's per-context writers do.  Python threads: the
    write(2) of a multi-megabyte buffer releases the interpreter lock, which is all the loop spends time in.  GB/s and files/s per count."""
    import threading
    out = {}
    buf = bytes(bytearray(os.urandom(4096)) * (max(int(nbytes), 4096) // 4096))
    for n in threads:
        errs = []

        def work(tid):
            try:
                for i in range(files_per_thread):
                    fn = os.path.join(directory, f"probe_{tid}_{i % 4}.bin")
                    with open(fn, "wb", buffering=0) as f:
                        f.write(buf)
            except OSError as e:
                errs.append(str(e))
        ts = [threading.Thread(target=work, args=(k,)) for k in range(n)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        for k in range(n):
            for i in range(4):
                try:
                    os.unlink(os.path.join(directory, f"probe_{k}_{i}.bin"))
                except OSError:
                    pass
        out[n] = {"error": errs[0]} if errs else {"GBs": n * files_per_thread * len(buf) / dt / 1e9, "files_per_s": n * files_per_thread / dt}
    return out


def host_block(bs, trees, buffers, bytes_per_frame, frames_per_s_per_gpu):
    """Where the host side of a delivered form lives and what it has to sustain PER GPU: the NUMA node of each context's GPU
    (bs_numa_node), the node its page-locked output buffers landed on (bs_host_page_node of their first and last page), and bytes/s."""
    from blackstar_amd import _lib
    L = _lib.lib()
    gpu_nodes = [t.numa_node() for t in trees]
    buf_nodes = []
    for ring in buffers:
        nodes = set()
        for b in ring:
            base, n = b.ctypes.data, b.nbytes
            nodes.update((int(L.bs_host_page_node(base)), int(L.bs_host_page_node(base + max(n - 1, 0)))))
        buf_nodes.append(sorted(nodes))
    return {"bytes_per_frame": int(bytes_per_frame), "GBs_needed_per_gpu": bytes_per_frame * frames_per_s_per_gpu / 1e9,
            "numa_node_gpu": gpu_nodes, "numa_node_buffers": buf_nodes,
            "buffers_on_gpu_node": all(g < 0 or nodes == [g] for g, nodes in zip(gpu_nodes, buf_nodes)),
            "threads": "one host thread per context inside the call, bound to the CPUs of the GPU's NUMA node by the library (bs::NumaBind; BLACKSTAR_NUMA_BIND=0 turns it off)"}


def predict_frames_8_gpus(frames_per_s_per_gpu, bytes_per_frame, host_GBs_per_gpu, host_GBs_total, what):
    """Frames are independent and every GPU renders whole frames with its own context, host thread, buffers (and writer): the GPU side of
    an 8-GPU run is 8x one GPU by construction.  What can fall short is the HOST side, which this measures on the box it runs on:
    host_GBs_per_gpu = what one GPU's delivery path sustains (its PCIe link / its writer thread), host_GBs_total = what the shared part
    sustains with 8 at once (None: nothing shared was measured).  predicted_speedup = min(8, per-GPU bound, shared bound) in units of
    one GPU's measured rate -- like split's prediction, a bound from one device, not a measurement of eight."""
    need = frames_per_s_per_gpu * bytes_per_frame / 1e9
    per_gpu = host_GBs_per_gpu / need if need > 0 and host_GBs_per_gpu else None          # x the rate one GPU needs
    shared = host_GBs_total / need if need > 0 and host_GBs_total else None              # GPUs' worth of the shared resource
    bound = 8.0
    if per_gpu is not None:
        bound = min(bound, 8.0 * min(1.0, per_gpu))
    if shared is not None:
        bound = min(bound, shared)
    return {"gpu_bound": 8.0, "GBs_needed_per_gpu": need, "host_GBs_per_gpu_measured": host_GBs_per_gpu, "host_GBs_shared_measured_8_at_once": host_GBs_total,
            "host_headroom_per_gpu": per_gpu, "host_bound": None if shared is None and per_gpu is None else (min(8.0 * min(1.0, per_gpu or 1e9), shared or 1e9)),
            "predicted_speedup": bound, "what": what}


def pcie_zero_copy_probe(bs, np, tree):
    """GB/s one GPU's kernels sustain writing a frame straight into page-locked host memory: BASELINE configs[1] (1920x1080, no
    supersampling: 49.8 MB of f64 from a 1.3 ms kernel -- the densest store stream this library produces) through bs_render into a
    bs_host_alloc buffer, best of 3 blocking calls.  A LOWER bound of the link (the kernel, not PCIe, sets the pace)."""
    cfg = workload_config(bs, "default").to_bs_config()
    buf = bs.alloc_image(tree, cfg["height"], cfg["width"])
    best = float("inf")
    for rep in range(4):
        t0 = time.perf_counter()
        bs.render(cfg, tree, out=buf)
        if rep:
            best = min(best, time.perf_counter() - t0)
    return buf.nbytes / best / 1e9


def d2h_forms(bs, np, trees, frame_objs, W, H, world, forms, fence, max_over_ranks, same_frames=False, all_ranks=None, split=None, extras=True):
    """The product's own batch entry points with every frame DELIVERED to the host (SURVEY 8d/7.6 "with and without D2H"):
    frame_objs[i] (a Config; its scene carries bloomStrength / bloomDivider) goes to trees[i % len(trees)].  Output buffers are
    page-locked (bs_host_alloc), a ring of 4 per context: two frames are in flight per context, so frame k's buffer is free again
    by the time frame k + 4 is enqueued -- the consumer (the reference writes each frame to a PNG file, app/Main.hs:121-123) is
    NOT part of the timed region.  Timed like the headline: fence, one blocking call, fence; max over ranks.
    same_frames: every frame_objs[i] is the same scene, so after the timed call (untimed) every delivered frame -- whichever context, device
    or rank made it -- must be byte-identical to the first: `frames_identical` (None when the frames differ by design: the animation).
    all_ranks(x) -> every rank's x; split() -> the `split` block (split_leg), supplied by the caller who knows the ranks."""
    n_t = len(trees)
    res = {}
    pcie = {}

    def pcie_GBs():   # measured once, on the first context, when the first form asks (untimed, after that form's own timed call)
        if "v" not in pcie:
            try:
                pcie["v"] = pcie_zero_copy_probe(bs, np, trees[0])
            except Exception as e:
                print(f"bench.py: PCIe probe failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                pcie["v"] = None
        return pcie["v"]

    def identical_everywhere(blobs):
        """blobs: this process's delivered frames (arrays or bytes).  All equal here, and -- through 48 bits of a digest -- on every rank.
        With ranks this is a collective: every rank calls it, also one that has nothing to show (which makes the answer False)."""
        if not same_frames:
            return None
        arrs = [np.frombuffer(b, np.uint8) if isinstance(b, (bytes, bytearray, memoryview)) else np.asarray(b) for b in blobs]
        here = bool(arrs) and all(np.array_equal(a, arrs[0]) for a in arrs[1:])
        if all_ranks is None:
            return here if arrs else None
        marks = all_ranks(digest_as_float(frame_digest(np, arrs[0])) if here else -1.0)
        return bool(min(marks) >= 0 and len(set(marks)) == 1)

    FORMS = {
        "batch": dict(key="batch", entry="bs_render_batch", call=lambda fo, o: bs.render_batch(fo, trees, outs=o),
                      alloc=lambda t: bs.alloc_image(t, H, W, dtype=np.float64), nbytes=lambda r: W * H * 3 * 8,
                      note="RGB f64 frames written into page-locked host memory by the trace kernels themselves (zero copy), two frames in flight per context"),
        "rgb8-batch": dict(key="rgb8_batch", entry="bs_render_rgb8_batch", call=lambda fo, o: bs.render_rgb8_batch(fo, trees, outs=o),
                           alloc=lambda t: bs.alloc_image(t, H, W, dtype=np.uint8), nbytes=lambda r: W * H * 3,
                           note="doRender on the device (render -> bloom -> sRGB8, app/Main.hs:105-123): only RGB8 reaches the host, two frames in flight per context"),
        "png-batch": dict(key="png_batch", entry="bs_render_png_batch", call=lambda fo, o: bs.render_png_batch(fo, trees, outs=o),
                          alloc=lambda t: bs.alloc_png(t, H, W), nbytes=lambda r: int(sum(len(f) for f in r) / max(len(r), 1)),
                          note="doRender on the device to the end (render -> bloom -> sRGB8 -> PNG encoder, app/Main.hs:105-123 with writeImg's file format): "
                               "the finished file is all that reaches the host (bytes_to_host_per_frame = its mean size); what is left for the host is write(2)"),
    }
    for form in forms:
        if form == "split":
            if split is not None:
                res["split"] = split()
            continue
        if form == "png-files":
            res["png_files"] = png_files_leg(bs, trees, frame_objs, W, H, world, fence, max_over_ranks, identical_everywhere, probe_host=extras)
            continue
        F = FORMS[form]
        # A rank on which anything of this form fails (page-locked memory, a device error) still goes through the SAME fences and collectives as
        # the others -- its time counts as infinite, its frames as absent -- and the form reports {"error"} on every rank instead of taking the
        # line (whose headline value is already measured) down with it or leaving the other ranks waiting in a barrier.
        err = None
        rings = outs = got = None
        call = F["call"]
        try:
            rings = [[F["alloc"](t) for _ in range(4)] for t in trees]
            outs = [rings[i % n_t][(i // n_t) % 4] for i in range(len(frame_objs))]
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        # untimed warm-up = the same call once, over at least WARM_PER_CONTEXT frames per context: the contexts' second stream, device images
        # and blur scratch get created, every ring buffer is written once, a one-off ~35 ms that the FIRST many-frame batch call of a process
        # pays when no other timed work preceded it (measured: 5.96 ms per frame in the first 20-frame call, 4.10-4.11 in the next three;
        # profiles/EXPERIMENTS.md) is spent -- and every context MEASURES this frame shape once (csrc/batch.cpp: the partition trial, 8 + 3 x 8
        # frames), so that the timed call runs with the context's remembered choice like any later call of a long-lived host would
        n_warm = max(len(frame_objs), WARM_PER_CONTEXT * n_t)
        if err is None:
            try:
                call([frame_objs[i % len(frame_objs)] for i in range(n_warm)], [rings[i % n_t][(i // n_t) % 4] for i in range(n_warm)])
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
        each = []
        for _ in range(TIMED_CALLS):   # the same blocking call TIMED_CALLS times, the faster one reported (20 frames are 90 ms: one host hiccup is 1 %)
            fence()
            t0 = time.perf_counter()
            if err is None:
                try:
                    got = call(frame_objs, outs)
                except Exception as e:
                    err = f"{type(e).__name__}: {e}"
            fence()
            each.append(max_over_ranks(float("inf") if err is not None else time.perf_counter() - t0))
        dt = min(each)
        if err is not None or max(each) == float("inf"):   # (every rank learns it through the max; the comparison below stays a collective for all)
            identical_everywhere([])
            print(f"bench.py: delivered form {form!r} failed{'' if err is None else ': ' + err}", file=sys.stderr, flush=True)
            res[F["key"]] = {"error": err or "another rank failed", "entry_point": F["entry"]}
            del rings, outs, got
            continue
        frames = len(frame_objs) * world // 1  # every rank runs the same number of frames
        per_gpu = len(frame_objs) / n_t
        res[F["key"]] = {
            "Mpixel_s": frames * W * H / dt / 1e6, "ms_per_frame_per_gpu": dt / per_gpu * 1e3, "frames": frames, "seconds": dt,
            "seconds_each_call": [round(t, 6) for t in each],
            "bytes_to_host_per_frame": F["nbytes"](got), "entry_point": F["entry"], "note": F["note"],
            # (ring buffers: the distinct ones hold the last frame written into each; PNG: every file of the call)
            "frames_identical": identical_everywhere([bytes(g) for g in got] if form == "png-batch" else list({id(o): o for o in outs}.values()))}
        if extras:   # where the host side lives and what an 8-GPU run asks of it (this rank's view; untimed)
            try:
                fps = per_gpu / dt
                nb = res[F["key"]]["bytes_to_host_per_frame"]
                res[F["key"]]["host"] = host_block(bs, trees, rings, nb, fps)
                res[F["key"]]["prediction_8_gpus"] = predict_frames_8_gpus(
                    fps, nb, pcie_GBs(), None,
                    "per GPU: its own PCIe link, measured as the rate this GPU's kernels write a frame into page-locked host memory (bs_render of BASELINE "
                    "configs[1], a lower bound of the link); shared: host DRAM takes 8x GBs_needed_per_gpu -- not measurable from one GPU, two sockets' "
                    "worth of DDR5 is several hundred GB/s")
            except Exception as e:
                res[F["key"]]["host"] = {"error": f"{type(e).__name__}: {e}"}
        del rings, outs, got
    return res


def png_files_leg(bs, trees, frame_objs, W, H, world, fence, max_over_ranks, identical_everywhere=None, probe_host=True):
    """The reference's batch loop to the very end: every frame rendered, bloomed, encoded AND written to a file by ONE bs_render_png_files
    call (per context: a rolling pipeline of frames in flight, a ring of page-locked file buffers and a native writer thread of its own),
    into a fresh directory on the RAM disk (or the temp directory) that is removed afterwards.  Untimed calls first, like the other
    delivered forms.  host = what every context's writer did in the timed call (bs_files_stats) and what this box's file system takes
    from 1 and from 8 writer threads at once in the same directory; prediction_8_gpus follows from those."""
    import shutil
    import tempfile
    # A directory with room for the files (a container's /dev/shm can be 64 MB): RAM disk if it has 4x what the frames need, else the temp
    # directory, else the leg is skipped -- on EVERY rank (the decision is a collective), so nobody waits in a fence for a rank that left.
    need = 4 * len(frame_objs) * (W * H * 3 + 4096)
    base = None
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            if os.path.isdir(cand) and os.access(cand, os.W_OK) and shutil.disk_usage(cand).free >= need:
                base = cand
                break
        except OSError:
            pass
    if max_over_ranks(0.0 if base else 1.0) > 0:
        return {"skipped": f"no directory with {need >> 20} MiB free on some rank (/dev/shm, {tempfile.gettempdir()})"}
    d = tempfile.mkdtemp(prefix="blackstar_bench_", dir=base)
    err, size, files, fstats, probe = None, 0, None, None, None
    try:
        paths = [os.path.join(d, f"f{i:05d}.png") for i in range(len(frame_objs))]
        # untimed warm-up over at least WARM_PER_CONTEXT frames per context (like d2h_forms): the partition trial (a warm-up segment plus
        # three 8-frame segments, 32-40 frames) must have ENDED before the timed call -- otherwise the timed region contains trial segments
        # (possibly a starved 8-CU one), not the remembered choice
        warm_calls = max(1, -(-WARM_PER_CONTEXT * len(trees) // max(len(frame_objs), 1)))
        try:
            for _ in range(warm_calls):
                bs.render_png_files(frame_objs, trees, paths)
        except Exception as e:  # a failing rank still goes through the same fences and collectives as the others
            err = f"{type(e).__name__}: {e}"
        each_local = []
        for rep in range(TIMED_CALLS):
            fence()
            t0 = time.perf_counter()
            if err is None:
                try:
                    bs.render_png_files(frame_objs, trees, paths)
                    each_local.append(time.perf_counter() - t0)
                    if each_local[-1] <= min(each_local):
                        fstats = [bs.files_stats(t) for t in trees] if hasattr(bs, "files_stats") else None
                except Exception as e:
                    err = f"{type(e).__name__}: {e}"
        if err is None:
            try:
                size = sum(os.path.getsize(p) for p in paths)
                if identical_everywhere is not None:  # read back before the directory goes (untimed; a few distinct files would do, all is simplest)
                    files = []
                    for p in paths:
                        with open(p, "rb") as f:
                            files.append(f.read())
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
        fence()
        if probe_host and err is None and size:   # untimed, same directory: one writer alone, and eight at once (what eight contexts' writers ask of this file system)
            try:
                probe = write_rate_probe(d, size // max(len(paths), 1))
            except Exception as e:
                probe = {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    # per call the slowest rank, then the faster call (every rank takes part in every collective, also one that failed: its times are infinite)
    each = [max_over_ranks(t) for t in (each_local if err is None and len(each_local) == TIMED_CALLS else [float("inf")] * TIMED_CALLS)]
    dt = min(each)
    same = identical_everywhere(files or []) if identical_everywhere is not None else None   # (a collective when there are ranks: every rank calls it)
    if err is not None or dt == float("inf"):
        return {"error": err or "another rank failed"}
    frames = len(frame_objs) * world
    per_gpu = len(frame_objs) / len(trees)
    out = {"Mpixel_s": frames * W * H / dt / 1e6, "ms_per_frame_per_gpu": dt / per_gpu * 1e3, "frames": frames, "seconds": dt,
           "seconds_each_call": [round(t, 6) for t in each],
           "frames_per_s": frames / dt, "bytes_written_per_frame": size // max(len(paths), 1), "entry_point": "bs_render_png_files",
           "directory": "RAM disk (/dev/shm)" if base == "/dev/shm" else base, "frames_identical": same, "warm_up_calls": warm_calls,
           "note": "scene to FILE: render -> bloom -> sRGB8 -> PNG encoder on the device; per context one rolling pipeline, a ring of page-locked file "
                   "buffers and a native writer thread of its own, no cross-context synchronisation (app/Main.hs:68-77 incl. writeImg's write)"}
    if fstats:
        nb = out["bytes_written_per_frame"]
        fps = per_gpu / dt
        out["host"] = {"writers": sum(s["writer_threads"] for s in fstats), "contexts": len(fstats), "ring_per_context": [s["ring"] for s in fstats],
                       "files_per_writer": [s["files"] for s in fstats],
                       "writer_busy_frac": max(s["writer_busy_frac"] for s in fstats), "writer_busy_frac_per_context": [round(s["writer_busy_frac"], 4) for s in fstats],
                       "buffer_wait_ms_per_context": [round(s["buffer_wait_ms"], 3) for s in fstats],
                       "bytes_per_frame": nb, "GBs_needed_per_gpu": nb * fps / 1e9,
                       "numa_node_gpu": [s["numa_node_gpu"] for s in fstats], "numa_node_buffers": [s["numa_node_buffers"] for s in fstats],
                       "threads_bound_to_gpu_node": [bool(s["threads_bound"]) for s in fstats],
                       "buffers_on_gpu_node": all(s["numa_node_gpu"] < 0 or s["numa_node_buffers"] == s["numa_node_gpu"] for s in fstats)}
        if isinstance(probe, dict) and all(isinstance(v, dict) and "GBs" in v for v in probe.values()) and probe:
            one, eight = probe.get(1), probe.get(8)
            out["host"]["one_thread_write_GBs"] = one["GBs"] if one else None
            out["host"]["eight_threads_write_GBs"] = eight["GBs"] if eight else None
            out["prediction_8_gpus"] = predict_frames_8_gpus(
                fps, nb, one["GBs"] if one else None, eight["GBs"] if eight else None,
                "per GPU: its own writer thread, measured as ONE thread creating / writing / closing files of this size in the target directory; shared: the "
                "file system under EIGHT such threads at once (same directory, same file size) -- both measured on this box after the timed call")
        elif probe is not None:
            out["host"]["write_probe"] = probe
    return out


def sustained_leg(bs, torch, np, trees, cfgs, outs, streams, devs, n_frames):
    """n_frames more frames per device, back to back on one stream each (all devices at once), with an event every 50 frames:
    what the chip sustains under its package power cap once it is warm, which 20 launches cannot show.  Returns per-device
    (ms per frame overall, first 50, last 50).  n_frames must be a multiple of 50."""
    assert n_frames >= 50 and n_frames % 50 == 0
    marks = []
    for k, d in enumerate(devs):
        with torch.cuda.device(d):
            marks.append([torch.cuda.Event(enable_timing=True) for _ in range(n_frames // 50 + 1)])
    t0 = time.perf_counter()
    for i in range(n_frames):
        for k, d in enumerate(devs):
            if i % 50 == 0:
                with torch.cuda.device(d):
                    marks[k][i // 50].record(streams[k])
            bs.render_device(cfgs[k], trees[k], outs[k].data_ptr(), outs[k].numel(), streams[k].cuda_stream)
    for k, d in enumerate(devs):
        with torch.cuda.device(d):
            marks[k][-1].record(streams[k])
    for d in sorted(set(devs)):
        torch.cuda.synchronize(d)
    wall = time.perf_counter() - t0
    per_dev = []
    for k in range(len(devs)):
        seg = [marks[k][j].elapsed_time(marks[k][j + 1]) / 50 for j in range(len(marks[k]) - 1)]  # n_frames is a multiple of 50
        per_dev.append({"ms_per_frame": marks[k][0].elapsed_time(marks[k][-1]) / n_frames, "ms_first_50": seg[0], "ms_last_50": seg[-1],
                        "ms_slowest_50": max(seg)})
    return wall, per_dev


def frame_digest(np, frame):
    """sha256 of a frame's bytes (a torch tensor resident on any device, or a numpy array).  Untimed: the image crosses PCIe once."""
    import hashlib
    if hasattr(frame, "detach"):
        t = frame.detach()
        if t.is_cuda:   # into PAGE-LOCKED host memory: a pageable destination makes the runtime pin pages on the fly (profiles/EXPERIMENTS.md section 5)
            import torch
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t)
            t = host
        a = t.numpy()
    else:
        a = np.asarray(frame)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digest_as_float(hexdigest):
    """48 bits of a digest as a float (exact in binary64): lets equality of frames be decided through the float collectives."""
    return float(int(hexdigest[:12], 16))


TIMED_CALLS = 2         # timed calls per delivered form (the faster one is the form's figure; both are listed)
WARM_PER_CONTEXT = 48   # frames per context of the delivered forms' untimed warm-up call (the partition trial needs 32, 40 on small frames)
COUNTERS = ("rays", "steps", "capped", "horizon", "escaped", "disk_hits", "star_hits")


def validation_block(per_device, what, repeat_digest=None):
    """What makes an N > 1 line prove itself: per_device = one (sha256 of the frame, bs_stats dict) per device /
    rank, all of the SAME frame rendered once more after the timed region.  The arithmetic is deterministic, so the frames must be
    bit-identical and every counter equal; a mismatch makes the line "valid": false (the measurement is still printed)."""
    digests = [d for d, _ in per_device]
    identical = len(set(digests)) == 1
    counters = {k: [int(st[k]) for _, st in per_device] for k in COUNTERS}
    counters_equal = all(len(set(v)) == 1 for v in counters.values())
    out = {"frame": what, "devices_compared": len(per_device), "frames_identical_across_devices": identical,
           "frame_sha256_per_device": [d[:16] for d in digests], "steps_per_device": counters["steps"],
           "counters_identical_across_devices": counters_equal,
           "counters_device0": {k: v[0] for k, v in counters.items()}}
    if not counters_equal:
        out["counters_per_device"] = counters
    valid = identical and counters_equal and all(v > 0 for v in counters["steps"])
    if repeat_digest is not None:  # the same frame twice on device 0: the kernel is deterministic run to run
        out["repeat_identical_on_device0"] = repeat_digest == digests[0]
        valid = valid and out["repeat_identical_on_device0"]
    out["valid"] = bool(valid)
    return out


def per_config_block(bs, torch, np, _lib, tree, args, device):
    """BASELINE configs[1] and configs[3] in the default line: per config, untimed launches for at least PER_CONFIG_WARM_MS
    (the headline's own ten warm-up launches are 43 ms; configs[1] gets a context of its own -- bs_create is host work during which the chip
    idles and drops its clocks, and two 1.3 ms launches do not bring them back: round 5's line read 0.64 where `--workload default` reads 0.71),
    then five launches each bracketed by HIP events on the launch stream, image resident in HBM like the headline.  frac = 145 flop x
    executed RK4 steps / mean launch time / the FP64 vector peak, exactly like roofline.frac.  About 0.3 s in all."""
    res = {}
    for name in ("default", "lensing-4k"):
        wl = WORKLOADS[name]
        cfg = workload_config(bs, name).to_bs_config()
        W, H = cfg["width"], cfg["height"]
        t = tree
        if not wl["stars"]:  # configs[1] has no star map: its own context, built from an empty star set
            t = bs.StarTree(None, device=device)
            t.set_mode(_lib.BS_MODE_FAST if args.mode == "fast" else _lib.BS_MODE_STRICT)
        try:
            img = torch.empty((H, W, 3), dtype=torch.float64, device=f"cuda:{device}")
            stream = torch.cuda.current_stream()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            t0 = time.perf_counter()
            for _ in range(2):
                bs.render_device(cfg, t, img.data_ptr(), img.numel(), stream.cuda_stream)
            torch.cuda.synchronize()
            est_ms = max((time.perf_counter() - t0) * 1e3 / 2, 0.05)     # (measured on cold clocks: up to 15 % high, which PER_CONFIG_WARM_MS allows for)
            for _ in range(min(200, int(PER_CONFIG_WARM_MS / est_ms) + 1)):
                bs.render_device(cfg, t, img.data_ptr(), img.numel(), stream.cuda_stream)
            for a, b in ev:
                a.record(stream)
                bs.render_device(cfg, t, img.data_ptr(), img.numel(), stream.cuda_stream)
                b.record(stream)
            torch.cuda.synchronize()
            st = t.stats()
            each = [float(a.elapsed_time(b)) for a, b in ev]
            ms = float(np.mean(each))
            executed = int(st["steps"]) - int(st["rays"])
            tf = FLOP_PER_STEP * executed / (ms * 1e-3) / 1e12
            res[name] = {"workload": wl["label"].format(cat="same catalogue as the headline"), "baseline_config": wl["baseline"],
                         "ms": ms, "ms_each": [round(x, 4) for x in each], "Mpixel_s": W * H / ms / 1e3, "rays": int(st["rays"]),
                         "steps": int(st["steps"]), "rk4_steps_executed": executed, "achieved_TFLOPs": tf, "frac": tf / PEAK_FP64_VALU_TFLOPS,
                         "lane_efficiency": st["steps"] / (64.0 * st["wave_iters"]), "effective_mode": ["strict", "fast"][int(st["effective_mode"])],
                         "frame_sha256": frame_digest(np, img)[:16]}
            del img
        finally:
            if t is not tree:
                t.close()
    return res


def predict_bands(bs, np, tree, cfg, H, W, n_bands=8, reps=2):
    """What an n_bands-GPU split of this frame will be limited by, measured on ONE device: each of the n_bands
    contiguous row bands bs_render_split would hand to a GPU is rendered alone (bs_render_rows into a page-locked band, like the split
    itself) and its executed steps, kernel time (hipEvent, bs_stats) and blocking-call time are recorded.  Two bounds follow: the WORK
    bound sum(steps) / (n * max band steps) -- what equal-height bands cost because central rows trace longer geodesics -- and the
    MEASURED bound sum(band call ms) / max(band call ms), which also carries the fixed cost per launch (each band pays the ~0.25 ms
    launch tail a whole frame pays once).  predicted_speedup_bound = one device's whole-frame call / the slowest band's call."""
    from blackstar_amd.distributed import shard_rows
    bands = [shard_rows(H, k, n_bands) for k in range(n_bands)]
    buf = bs.alloc_image(tree, max(b - a for a, b in bands), W)
    full = bs.alloc_image(tree, H, W)
    for _ in range(3):   # the chip's clocks are up before anything is timed (they drop within a millisecond of idling and need ~8 launches / 35 ms to come back)
        bs.render(cfg, tree, out=full)
    # passes over all bands, the FASTEST pass of each band kept: a band is a 2-3 ms launch, so a clock ramp or a host hiccup is a large
    # relative error, and it is the steady state an 8-GPU run sees that is being predicted
    steps = [0] * n_bands
    kernel_ms, call_ms = [float("inf")] * n_bands, [float("inf")] * n_bands
    for rep in range(reps + 1):   # (pass 0 untimed: buffers touched)
        for k, (a, b) in enumerate(bands):
            out = buf[: b - a]
            t0 = time.perf_counter()
            bs.render_rows(cfg, tree, a, b, out=out)
            dt = (time.perf_counter() - t0) * 1e3
            st = tree.stats()
            steps[k] = int(st["steps"])
            if rep:
                kernel_ms[k] = min(kernel_ms[k], float(st["kernel_ms"]))
                call_ms[k] = min(call_ms[k], dt)
    whole, whole_k = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        bs.render(cfg, tree, out=full)
        whole.append((time.perf_counter() - t0) * 1e3)
        whole_k.append(float(tree.stats()["kernel_ms"]))
    whole_ms, whole_kernel = float(min(whole)), float(min(whole_k))
    tot = float(sum(steps))
    return {"n_bands": n_bands, "bands": [list(b) for b in bands], "band_steps": steps, "band_kernel_ms": [round(x, 4) for x in kernel_ms],
            "band_call_ms": [round(x, 4) for x in call_ms], "whole_frame_call_ms": whole_ms, "whole_frame_kernel_ms": whole_kernel,
            "steps_max_over_mean": max(steps) / (tot / n_bands) if tot else None,
            "work_bound": tot / max(steps) if tot else None,
            "kernel_bound": whole_kernel / max(kernel_ms) if max(kernel_ms) > 0 else None,
            "predicted_speedup_bound": whole_ms / max(call_ms) if max(call_ms) > 0 else None,
            "fixed_ms_per_band": (sum(kernel_ms) - whole_kernel) / n_bands,
            "note": f"every one of the {n_bands} row bands an {n_bands}-GPU bs_render_split cuts, rendered alone on ONE device after a warm-up (fastest of {reps} passes): "
                    "work_bound = sum(steps) / max(band steps) (equal-height bands, central rows cost more); kernel_bound and "
                    "predicted_speedup_bound = the whole frame on one device / the slowest band (kernel time; blocking call) -- the latter two "
                    "include the fixed cost every launch pays (fixed_ms_per_band = (sum of band kernels - whole-frame kernel) / bands)"}


def split_leg(bs, np, trees, rank, world, fence, max_over_ranks, gather_objs, reps=3):
    """ONE frame of BASELINE configs[3] (lensing-disk at 3840x2160, 4x supersample) cut into row bands over all GPUs -- SURVEY 8e's
    fallback for a single huge frame.  One process with N contexts: bs_render_split (one host thread per context, every GPU writes its
    band straight into the caller's page-locked frame).  One process per GPU: each rank renders band `rank` with bs_render_rows.  Either
    way the bands are compared byte for byte with the whole frame rendered by ONE device (untimed), and the one-device time of the
    same call is reported beside it: total work is fixed, so this is the strong-scaling figure."""
    from blackstar_amd.distributed import shard_rows
    cfg_obj = workload_config(bs, "lensing-4k")
    cfg = cfg_obj.to_bs_config()
    W, H = cfg["width"], cfg["height"]
    n_t = len(trees)
    n_parts = n_t * world
    errs = []   # a rank on which a render fails keeps taking part in every fence and collective (its times count as infinite): nobody hangs

    def alloc(rows):
        try:
            return bs.alloc_image(trees[0], rows, W)
        except Exception as e:   # (page-locked memory: the placeholder is never rendered into -- every later render is skipped once errs is set)
            errs.append(f"{type(e).__name__}: {e}")
            return np.zeros((1, W, 3))

    ref = alloc(H)

    def guarded(fn):
        if errs:
            return
        try:
            fn()
        except Exception as e:
            errs.append(f"{type(e).__name__}: {e}")

    def timed(fn):
        guarded(fn)  # untimed: buffers touched, streams made
        ts = []
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            guarded(fn)
            fence()
            ts.append(max_over_ranks(float("inf") if errs else time.perf_counter() - t0))
        return ts

    one = timed(lambda: bs.render(cfg, trees[0], out=ref))   # the whole frame on one device (every rank does this: its own reference)
    st1 = {"steps": 0, "rays": 0, "wave_iters": 0, "kernel_ms": 0.0}
    guarded(lambda: st1.update(trees[0].stats()))
    ref_steps = int(st1["steps"])
    if world == 1:
        full = alloc(H)
        full[:] = 0
        ts = timed(lambda: bs.render_split(cfg, trees, out=full))
        identical = not errs and bool(np.array_equal(full, ref))
        bands = [shard_rows(H, k, n_t) for k in range(n_t)]
        entry = "bs_render_split"
    else:
        row0, row1 = shard_rows(H, rank, world)
        band = alloc(row1 - row0)
        band[:] = 0
        ts = timed(lambda: bs.render_rows(cfg, trees[0], row0, row1, out=band))
        mine = not errs and bool(np.array_equal(band, ref[row0:row1]))   # this rank's band against this rank's own whole frame ...
        got = gather_objs((mine, frame_digest(np, ref) if not errs else f"rank {rank} failed: {errs[0]}", (row0, row1)))
        identical = all(g[0] for g in got) and len({g[1] for g in got}) == 1   # ... and every rank's whole frame is the same frame
        bands = [g[2] for g in got]
        entry = "bs_render_rows (one band per rank)"
    dt, dt_one = float(np.mean(ts)), float(np.mean(one))
    if errs or not np.isfinite(dt) or not np.isfinite(dt_one):   # (an infinite time = a rank failed; every rank sees it through the max)
        print(f"bench.py: split leg failed{': ' + errs[0] if errs else ' on another rank'}", file=sys.stderr, flush=True)
        return {"error": errs[0] if errs else "another rank failed", "entry_point": entry, "parts": n_parts, "identical_to_one_device": None}
    try:   # (after the timed calls; one device, no collective: every rank may do it, rank 0's is printed)
        prediction = predict_bands(bs, np, trees[0], cfg, H, W, 8, reps=3) if rank == 0 else None
    except Exception as e:
        prediction = {"error": f"{type(e).__name__}: {e}"}
    return {"prediction_8_gpus": prediction, "Mpixel_s": W * H / dt / 1e6, "ms_per_frame": dt * 1e3, "ms_each": [round(t * 1e3, 4) for t in ts], "seconds": dt, "frames": 1,
            "parts": n_parts, "bands": [list(b) for b in bands], "entry_point": entry,
            "one_device_ms_per_frame": dt_one * 1e3, "one_device_Mpixel_s": W * H / dt_one / 1e6, "speedup_vs_one_device": dt_one / dt,
            "identical_to_one_device": identical, "one_device_steps": ref_steps, "bytes_to_host_per_frame": W * H * 24,
            "one_device_stats": {k: (float(st1[k]) if k == "kernel_ms" else int(st1[k])) for k in ("rays", "steps", "wave_iters", "kernel_ms")},
            "workload": WORKLOADS["lensing-4k"]["label"].format(cat="same catalogue as the headline"),
            "note": "ONE frame, row bands over all GPUs, RGB f64 written into page-locked host memory by the kernels themselves; blocking call, "
                    f"mean of {reps}; strong scaling: speedup_vs_one_device is the figure for a single huge frame (SURVEY 8e)"}


def optional_leg(name, safe, fn):
    """Run a leg that is reported NEXT TO the headline value.  With safe=True (one rank, and the headline does not come from this
    leg) a failure becomes {"error": ...} in the line instead of costing the value already measured; otherwise it propagates
    (with several ranks, one rank leaving a leg early would leave the others waiting in its barrier)."""
    if not safe:
        return fn()
    try:
        return fn()
    except Exception as e:
        print(f"bench.py: optional leg {name!r} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        return {"error": f"{type(e).__name__}: {e}"}


def roofline_block(args, st, kernel_ms, W, H, peak_measured=None):
    executed = int(st["steps"]) - int(st["rays"])  # the kernel skips the reference's final, discarded rk4 per ray
    flops = FLOP_PER_STEP * executed
    achieved = flops / (kernel_ms * 1e-3) / 1e12  # mean launch duration over the timed region (HIP events on the launch stream)
    alg_bytes = 24.0 * W * H
    live = getattr(args, "traffic_live", None)
    if args.traffic_bytes is not None:
        traffic, traffic_src, kind = args.traffic_bytes, "--traffic-bytes", "given"
    elif live and live[0] is not None:
        traffic, traffic_src, kind = live[0], live[1], "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over scripts/prof_frame.py, same frame, child processes"
    else:
        traffic, traffic_src = pmc_traffic(args.mode)
        kind = "static (committed rocprofv3 --pmc passes, not measured in this run" + (f"; live measurement unavailable: {live[1]})" if live else ")")
    r = {"bound": "valu", "detail": "FP64 VALU issue (scalar ODE per lane; HBM and MFMA are not the bound)",
         "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS,
         "flop_kind": "reference-equivalent: 145 flop per RK4 step as the reference's arithmetic counts them (SURVEY 8d), "
                      "NOT executed instructions -- see valu_issue_frac for those",
         "flop_per_launch": flops, "flop_per_step": FLOP_PER_STEP, "rk4_steps_executed": executed,
         "traffic": traffic, "traffic_kind": kind,
         "traffic_source": traffic_src,
         "hbm": {"algorithmic_bytes": alg_bytes, "achieved_GBs": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                 "peak_GBs": PEAK_HBM_GBS, "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}}
    if live and live[0] is not None and isinstance(live[1], dict) and live[1].get("GRBM_GUI_ACTIVE"):
        r.update(cycles_view(flops, live[1]["GRBM_GUI_ACTIVE"], kernel_ms))
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
        _lib.check(_lib.debug_lib().bs_debug_ubench(tree.handle, 0, 256 * 8, 20000, C.byref(ms), C.byref(gi)), "bs_debug_ubench")
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


def label_roofline_scope(res, world, per_rank_ms):
    """Whose launches `roofline` describes.  N = 1: the one GPU's.  N > 1: RANK 0's (its HIP events, its bs_stats) -- a per-GPU figure, not
    an aggregate -- with every rank's own fraction beside it, derived from that rank's step time (per_rank_ms_per_step; the frames have
    the same flop count on every rank for a single-frame workload): min / mean / max show a straggler that rank 0's figure would hide."""
    r = res.get("roofline")
    if not isinstance(r, dict):
        return
    if world == 1:
        r["scope"] = "the one GPU of this run"
        return
    r["scope"] = "per GPU, rank 0 (device 0): kernel time, counters and flop count are rank 0's; value / ms_per_step are the whole job's"
    fl = r.get("flop_per_launch")
    ms = [m for m in (per_rank_ms or []) if m and m > 0 and m != float("inf")]
    if fl and ms and res.get("config", {}).get("launches_in_flight_per_gpu", 1) == 1:
        fr = [fl / (m * 1e-3) / 1e12 / r["peak"] for m in ms]
        r["frac_per_rank_from_step_time"] = {"min": min(fr), "mean": sum(fr) / len(fr), "max": max(fr), "ranks": len(fr),
                                             "note": "flop_per_launch / that rank's ms per step / peak: includes the few-microsecond gap between launches, so slightly below frac"}


def devices_or_die(world, ndev, allow_smoke):
    """--gpus N on a box that shows SOME but not ALL of the devices (2 <= visible < N) is a broken lease, not a smoke run: refuse loudly
    instead of putting several ranks on one device and printing a line that looks like a result.  One visible device is the documented
    smoke mode (`oversubscribed: true`); BLACKSTAR_BENCH_ALLOW_OVERSUBSCRIBE=1 allows the in-between case for experiments."""
    if world <= ndev or ndev == 1 or allow_smoke:
        return
    raise SystemExit(f"bench.py --gpus {world}: only {ndev} HIP devices are visible.  A device is missing (check the lease / HIP_VISIBLE_DEVICES / "
                     f"ROCR_VISIBLE_DEVICES); refusing to oversubscribe {ndev} devices with {world} ranks.  (One visible device = the smoke mode; "
                     f"BLACKSTAR_BENCH_ALLOW_OVERSUBSCRIBE=1 overrides.)")


def forms_valid(d2h):
    """False if any delivered form of this line found frames that should be identical and are not."""
    if not isinstance(d2h, dict):
        return True
    return all(v.get("frames_identical") is not False and v.get("identical_to_one_device") is not False for v in d2h.values() if isinstance(v, dict))


def legs_failed(d2h):
    """The delivered forms / split leg of this line that ended in an error on some rank (a product entry point failed on a device): the
    measurement of the other legs stands, but a reader of `valid` alone must not miss it -- bench.py puts the list in the line and makes
    the line invalid."""
    if not isinstance(d2h, dict):
        return []
    if "error" in d2h and not any(isinstance(v, dict) for v in d2h.values()):
        return ["with_d2h"]
    return sorted(k for k, v in d2h.items() if isinstance(v, dict) and "error" in v)


def split_headline(args, res, blk, world):
    """--form split: ONE frame over all GPUs is the result.  The line keeps the contract's keys, read for one frame: steps = 1 frame,
    ms_per_step = its wall time, scaling = strong (total work fixed as N grows)."""
    wl = WORKLOADS["lensing-4k"]
    res.update({"metric": wl["metric"] + ", ONE frame split by rows over all GPUs", "scaling": "strong", "steps": 1, "ms_per_step": blk["ms_per_frame"]})
    res["config"].update({"workload": blk["workload"], "baseline_config": wl["baseline"], "parallelism": f"row bands x{blk['parts']}",
                          "frames_per_step_per_gpu": f"1/{blk['parts']}", "image": blk["note"]})
    res["split"] = {k: blk[k] for k in ("speedup_vs_one_device", "one_device_ms_per_frame", "identical_to_one_device", "bands", "entry_point")}
    for k in ("rays_per_s", "steps_per_ray", "lane_efficiency", "kernel_ms", "kernel_ms_last_hipevent"):
        res.pop(k, None)   # they describe the warm-up launches' statistics, not the split frame
    st1 = blk["one_device_stats"]   # the roofline of this frame: its kernel on ONE device, whole frame (hipEvent time from bs_stats)
    res["roofline"] = dict(roofline_block(args, st1, st1["kernel_ms"], 3840, 2160), time_basis="the whole frame's kernel on one device (bs_stats.kernel_ms, HIP events)")


def result_line(args, world, launcher, value, dt, W, H, frames_cfg, st, kernel_ms, extra_cfg=None, peak_measured=None, cat_note=CATALOGUES["synthetic"]):
    overlapped = bool(extra_cfg) and extra_cfg.get("launches_in_flight_per_gpu", 1) > 1
    wl = WORKLOADS[getattr(args, "workload", "default-aa") if frames_cfg is None else "animation"]
    cfgd = {"workload": wl["label"].format(cat=cat_note), "baseline_config": wl["baseline"], "mode": args.mode, "frames_per_step_per_gpu": 1,
            "parallelism": f"frame-sharded x{world}", "launcher": launcher,
            "image": "RGB f64 resident in HBM (no D2H in the timed region)"}
    if extra_cfg:
        cfgd.update(extra_cfg)
    frames = args.steps * world
    return {
        "metric": wl["metric"], "value": value, "unit": "Mpixel/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if args.catalogue in CATALOGUES else "synthetic scene + real catalogue file",
        "config": cfgd,
        "rays_per_s": frames * st["rays"] / dt, "steps_per_ray": st["steps"] / st["rays"],
        "lane_efficiency": st["steps"] / (64.0 * st["wave_iters"]),
        "kernel_ms": kernel_ms, "kernel_ms_last_hipevent": st["kernel_ms"],
        # launches in flight overlap: a launch's own duration then says nothing about the rate; the step time does
        "roofline": dict(roofline_block(args, st, kernel_ms if not overlapped else dt / args.steps * 1e3, W, H, peak_measured),
                         time_basis="mean launch duration (HIP events)" if not overlapped else
                         "ms_per_step (launches overlap: two in flight per GPU; their own durations are about twice this)"),
    }
