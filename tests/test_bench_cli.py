"""CPU tests of bench.py's command-line contract and of its host-side helpers (no GPU: the legs themselves run on the MI355X box)."""
import json
import os
import subprocess
import sys
import time
import types

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_help_and_defaults():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--form", "--catalogue", "--sustained-frames", "--launcher", "--gather", "--workload"):
        assert flag in r.stdout, flag
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse_args()
    finally:
        sys.argv = old
    assert (a.gpus, a.steps, a.warmup, a.form, a.catalogue, a.mode, a.workload) == (1, 20, 10, "all", "synthetic", "fast", "default-aa")
    assert a.sustained_frames == 300 and a.cpu_seconds > 0


def test_no_gpu_means_no_result_not_a_fallback():
    """Without a HIP device bench.py must refuse (exit status != 0, no JSON line): never a CPU number dressed as the metric."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for extra in ([], ["--gpus", "2"]):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0"] + extra, capture_output=True, text=True, env=env)
        assert r.returncode != 0 and "needs a HIP device" in r.stderr + r.stdout
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_result_line_carries_the_contract_keys():
    args = types.SimpleNamespace(steps=20, warmup=3, mode="fast", traffic_bytes=None, catalogue="synthetic")
    st = {"steps": 1854063332, "rays": 8294400, "wave_iters": 29008800, "kernel_ms": 4.4}
    res = bench.result_line(args, 1, "single-process", 470.0, 0.0882, 1920, 1080, None, st, 4.41)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in res, k
    assert res["unit"] == "Mpixel/s" and res["dtype"] == "f64" and res["scaling"] == "weak" and res["vs_baseline"] is None and res["data"] == "synthetic"
    assert "workload" in res["config"] and "model" not in res["config"] and "BASELINE configs[2]" in res["config"]["workload"]
    r = res["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 145 * (st["steps"] - st["rays"]) / 4.41e-3 / 1e12) < 1e-9
    json.dumps(res)
    args.catalogue = "/data/ppm.cat"
    res = bench.result_line(args, 1, "single-process", 470.0, 0.0882, 1920, 1080, None, st, 4.41, cat_note=bench.catalogue_note(args, 378910))
    assert "REAL catalogue file ppm.cat" in res["config"]["workload"] and "reported separately" in res["config"]["workload"] and res["data"] != "synthetic"


def test_d2h_forms_drive_the_products_batch_entry_points_with_a_ring_of_four_buffers_per_context(monkeypatch):
    """with_d2h: frame i goes to context i % N through ONE bs_render_batch / bs_render_rgb8_batch call; two frames are in flight per
    context, so a ring of 4 page-locked buffers per context is enough; the warm-up touches every ring buffer once.  Every form says where
    its host side lives (`host`: NUMA node of GPU and buffers, GB/s needed per GPU; png_files: what each context's writer did and what
    1 and 8 writer threads get out of the target directory) and what that allows at 8 GPUs (`prediction_8_gpus`)."""
    calls = []

    class FakeTree:
        def numa_node(self):
            return 0

    class FakeBs:
        @staticmethod
        def alloc_image(tree, h, w, dtype=np.float64):
            return np.zeros((h, w, 3), dtype)

        @staticmethod
        def render_batch(cfgs, trees, outs=None):
            calls.append(("batch", len(cfgs), len(trees), [id(o) for o in outs], outs[0].dtype))
            time.sleep(0.002)
            return outs

        @staticmethod
        def render_rgb8_batch(cfgs, trees, outs=None):
            calls.append(("rgb8", len(cfgs), len(trees), [id(o) for o in outs], outs[0].dtype))
            time.sleep(0.001)
            return outs

        @staticmethod
        def render_rgb8(cfg, tree):
            return (np.arange(8 * 16 * 3) % 7).astype(np.uint8).reshape(8, 16, 3)

        @staticmethod
        def render_png_files(cfgs, trees, paths, pipe=16):
            calls.append(("files", len(cfgs), len(trees), [], None))
            for p in paths:
                with open(p, "wb") as f:
                    f.write(b"x" * 77)

        @staticmethod
        def files_stats(tree):
            return {"files": 10, "bytes": 770, "wall_ms": 8.0, "writer_busy_ms": 2.0, "buffer_wait_ms": 0.0, "ring": 10, "writer_threads": 1,
                    "numa_node_gpu": 0, "numa_node_buffers": 0, "threads_bound": 1, "writer_busy_frac": 0.25}

        @staticmethod
        def alloc_png(tree, h, w):
            return np.zeros(h * w * 3 + 100, np.uint8)

        @staticmethod
        def render_png_batch(cfgs, trees, outs=None):
            calls.append(("png", len(cfgs), len(trees), [id(o) for o in outs], outs[0].dtype))
            return [memoryview(o)[:40 + i % 2] for i, o in enumerate(outs)]   # files of 40 and 41 bytes

    trees = [FakeTree(), FakeTree(), FakeTree()]
    frames = [f"cfg{i}" for i in range(30)]
    fences = []
    res = bench.d2h_forms(FakeBs, np, trees, frames, 16, 8, 1, ["batch", "rgb8-batch", "png-batch", "png-files"], lambda: fences.append(1), lambda x: x)
    assert set(res) == {"batch", "rgb8_batch", "png_batch", "png_files"}
    assert bench.TIMED_CALLS == 2 and len(fences) == 3 * 4 + 3     # a fence either side of each timed call; png-files: one before each call, one after the last
    W = bench.WARM_PER_CONTEXT * 3   # the warm-up call is long enough for every context to measure the frame shape (the partition trial)
    assert W >= 32 * 3
    assert [c[:3] for c in calls] == [("batch", W, 3), ("batch", 30, 3), ("batch", 30, 3), ("rgb8", W, 3), ("rgb8", 30, 3), ("rgb8", 30, 3),
                                      ("png", W, 3), ("png", 30, 3), ("png", 30, 3),
                                      ] + [("files", 30, 3)] * 7  # warm-up call(s), two timed calls (the faster is the figure); png-files warms up 5 x 30 >= W frames too
    assert all(len(res[k]["seconds_each_call"]) == 2 and abs(res[k]["seconds"] - min(res[k]["seconds_each_call"])) < 1e-6 for k in res)
    assert res["png_files"]["warm_up_calls"] == 5
    assert res["png_files"]["bytes_written_per_frame"] == 77 and res["png_files"]["frames"] == 30 and res["png_files"]["entry_point"] == "bs_render_png_files"
    del calls[9:]
    assert res["png_batch"]["entry_point"] == "bs_render_png_batch" and res["png_batch"]["bytes_to_host_per_frame"] == 40   # the mean file size
    assert calls[1][4] == np.float64 and calls[4][4] == np.uint8
    ids = calls[1][3]
    assert len(set(ids)) == 12                                   # 4 buffers x 3 contexts
    for i in range(30):
        assert ids[i] == ids[(i % 3) + 3 * ((i // 3) % 4)]        # frame i: context i % 3, ring slot (i // 3) % 4
        for j in range(i + 1, min(i + 6, 30)):                   # frames that can be in flight together never share a buffer
            assert ids[i] != ids[j] or (j - i) >= 12
    assert set(calls[0][3]) == set(ids)                          # the warm-up wrote every ring buffer
    b = res["batch"]
    assert b["frames"] == 30 and b["entry_point"] == "bs_render_batch" and b["bytes_to_host_per_frame"] == 16 * 8 * 24
    assert abs(b["Mpixel_s"] - 30 * 16 * 8 / b["seconds"] / 1e6) < 1e-9 and abs(b["ms_per_frame_per_gpu"] - b["seconds"] / 10 * 1e3) < 1e-9
    assert res["rgb8_batch"]["bytes_to_host_per_frame"] == 16 * 8 * 3
    for key in ("batch", "rgb8_batch", "png_batch"):
        h, pr = res[key]["host"], res[key]["prediction_8_gpus"]
        assert h["numa_node_gpu"] == [0, 0, 0] and len(h["numa_node_buffers"]) == 3 and h["bytes_per_frame"] == res[key]["bytes_to_host_per_frame"]
        assert abs(h["GBs_needed_per_gpu"] - h["bytes_per_frame"] * 10 / res[key]["seconds"] / 1e9) < 1e-12
        assert pr["gpu_bound"] == 8.0 and pr["predicted_speedup"] == 8.0 and pr["host_GBs_per_gpu_measured"] is None   # (no PCIe probe without a GPU: nothing measured, nothing claimed)
    f = res["png_files"]
    assert f["host"]["writers"] == 3 and f["host"]["writer_busy_frac"] == 0.25 and f["host"]["files_per_writer"] == [10, 10, 10]
    assert f["host"]["buffers_on_gpu_node"] and f["host"]["threads_bound_to_gpu_node"] == [True] * 3
    assert f["host"]["one_thread_write_GBs"] > 0 and f["host"]["eight_threads_write_GBs"] > 0
    pr = f["prediction_8_gpus"]
    assert 0 < pr["predicted_speedup"] <= 8.0 and pr["host_GBs_per_gpu_measured"] == f["host"]["one_thread_write_GBs"]
    # the rule itself: 8x one GPU unless the host's per-GPU path or its shared part falls short
    p = bench.predict_frames_8_gpus(200.0, 2.5e6, 5.0, 40.0, "x")      # 0.5 GB/s needed per GPU: one writer does 10x, eight together 80 GPUs' worth
    assert p["predicted_speedup"] == 8.0 and abs(p["host_headroom_per_gpu"] - 10.0) < 1e-9
    assert bench.predict_frames_8_gpus(200.0, 2.5e6, 5.0, 2.0, "x")["predicted_speedup"] == 4.0      # the shared part carries 4 GPUs' worth
    assert bench.predict_frames_8_gpus(200.0, 2.5e6, 0.25, None, "x")["predicted_speedup"] == 4.0    # each GPU's own path does half of what it needs


def test_png_files_leg_skips_on_every_rank_when_there_is_no_room(monkeypatch):
    """A container's /dev/shm can be 64 MB: without room for the files the leg is skipped -- after the collective that makes the decision
    the same on every rank, before any fence -- and a write that fails later is reported, not raised (ranks would wait for each other)."""
    import collections
    import shutil
    collectives, fences = [], []
    usage = collections.namedtuple("usage", "total used free")
    monkeypatch.setattr(shutil, "disk_usage", lambda p: usage(1 << 30, 1 << 30, 1 << 20))

    class Bs:
        @staticmethod
        def render_png_files(cfgs, trees, paths, pipe=16):
            raise AssertionError("must not render")

    r = bench.png_files_leg(Bs, ["t"], ["c"] * 20, 1920, 1080, 2, lambda: fences.append(1), lambda x: (collectives.append(x), x)[1])
    assert "skipped" in r and collectives == [1.0] and fences == []
    monkeypatch.setattr(shutil, "disk_usage", lambda p: usage(1 << 40, 0, 1 << 40))

    class Failing:
        @staticmethod
        def render_png_files(cfgs, trees, paths, pipe=16):
            raise OSError("disk full")

    collectives.clear()
    r = bench.png_files_leg(Failing, ["t"], ["c"] * 4, 16, 8, 2, lambda: fences.append(1), lambda x: (collectives.append(x), x)[1])
    assert "disk full" in r["error"] and len(fences) == 3 and collectives == [0.0, float("inf"), float("inf")]   # same fences and collectives as a healthy rank (two timed calls)


def test_device_sampler_reads_only_the_devices_it_was_given(tmp_path):
    """The box's sysfs lists every GPU of the host (other tenants' too): devices are matched by PCI bus, unmatched ones are not reported."""
    for card, bus, clk, pw in ((0, "75", 2403000000, 300000000), (24, "26", 2100000000, 1250000000)):
        dev = tmp_path / "devices" / f"0000:{bus}:00.0"
        hw = dev / "hwmon" / "hwmon3"
        hw.mkdir(parents=True)
        (hw / "freq1_input").write_text(f"{clk}\n")
        (hw / "power1_input").write_text(f"{pw}\n")
        os.makedirs(tmp_path / "drm" / f"card{card}")
        os.symlink(dev, tmp_path / "drm" / f"card{card}" / "device")
    with bench.DeviceSampler([0x26], sysfs=str(tmp_path / "drm")) as s:
        time.sleep(0.15)
    out = s.summary()
    assert len(out) == 1 and out[0]["pci_bus"] == "26" and out[0]["samples"] >= 2
    assert out[0]["sclk_MHz_mean"] == 2100.0 and out[0]["power_W_mean"] == 1250.0
    with bench.DeviceSampler([0x99, None], sysfs=str(tmp_path / "drm")) as s:
        time.sleep(0.05)
    assert s.summary() is None


def test_catalogue_option_reads_a_real_ppm_file(tmp_path):
    """--catalogue PATH: the bytes of a PPM catalogue file go through the product's own reader (src/StarMap.hs:45-58 layout)."""
    import blackstar_amd as bs
    from blackstar_amd import synthetic
    data = synthetic.ppm_catalogue_bytes(300, seed=5)
    path = tmp_path / "PPM"
    path.write_bytes(data)
    assert synthetic.catalogue_bytes(str(path)) == data
    assert len(bs.read_map(synthetic.catalogue_bytes(str(path)))) == 300
    assert synthetic.catalogue_bytes("synthetic") == synthetic.ppm_catalogue_bytes()
    assert len(synthetic.catalogue_bytes("clustered")) > len(synthetic.catalogue_bytes("synthetic"))
    args = types.SimpleNamespace(catalogue=str(path))
    assert "REAL catalogue file PPM (300 stars" in bench.catalogue_note(args, 300)


def test_workloads_cover_every_single_gpu_baseline_config():
    """--workload names BASELINE configs[1], [2], [3] (and [4], the animation); each resolves to the reference's scene file with BASELINE's
    resolution override, and the result line names the config it measured."""
    import blackstar_amd as bs
    from oracle import scenes
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    for w in ("default-aa", "default", "lensing-4k", "animation", "split", "--no-validate"):
        assert w in r.stdout, w
    want = {"default": scenes.DEFAULT, "default-aa": scenes.DEFAULT_AA, "lensing-4k": scenes.with_res(scenes.LENSING_DISK, 3840, 2160)}
    for name, ref in want.items():
        got = bench.workload_config(bs, name).to_bs_config()
        for k, v in ref.items():
            assert np.allclose(got[k], v, rtol=0, atol=1e-15), (name, k, got[k], v)
    assert [bench.WORKLOADS[w]["baseline"] for w in ("default", "default-aa", "lensing-4k", "animation")] == ["configs[1]", "configs[2]", "configs[3]", "configs[4]"]
    assert not bench.WORKLOADS["default"]["stars"] and bench.WORKLOADS["lensing-4k"]["stars"]
    st = {"steps": 1854063332, "rays": 8294400, "wave_iters": 29008800, "kernel_ms": 4.4}
    for name in ("default", "lensing-4k"):
        args = types.SimpleNamespace(steps=20, warmup=3, mode="fast", traffic_bytes=None, catalogue="synthetic", workload=name)
        res = bench.result_line(args, 1, "single-process", 470.0, 0.0882, 1920, 1080, None, st, 4.41)
        assert res["config"]["baseline_config"] == bench.WORKLOADS[name]["baseline"] and bench.WORKLOADS[name]["baseline"] in res["config"]["workload"]
        assert res["metric"] == bench.WORKLOADS[name]["metric"] and "default-aa" not in res["metric"]


def test_validation_block_tells_a_right_answer_from_a_wrong_one():
    """The N > 1 line proves itself: same frame on every device -> same sha256, same counters; anything else is "valid": false."""
    st = {"rays": 8294400, "steps": 1854063332, "capped": 0, "horizon": 350000, "escaped": 7944400, "disk_hits": 1700000, "star_hits": 9000000}
    good = bench.validation_block([("ab" * 32, dict(st))] * 8, "frame", repeat_digest="ab" * 32)
    assert good["valid"] and good["frames_identical_across_devices"] and good["devices_compared"] == 8 and good["steps_per_device"] == [st["steps"]] * 8
    assert good["repeat_identical_on_device0"] and good["counters_identical_across_devices"] and "counters_per_device" not in good
    bad_frame = bench.validation_block([("ab" * 32, dict(st))] * 7 + [("cd" * 32, dict(st))], "frame")
    assert not bad_frame["valid"] and not bad_frame["frames_identical_across_devices"] and bad_frame["counters_identical_across_devices"]
    bad_steps = bench.validation_block([("ab" * 32, dict(st)), ("ab" * 32, dict(st, steps=st["steps"] - 1))], "frame")
    assert not bad_steps["valid"] and bad_steps["frames_identical_across_devices"] and not bad_steps["counters_identical_across_devices"]
    assert bad_steps["counters_per_device"]["steps"] == [st["steps"], st["steps"] - 1]
    idle = bench.validation_block([("ab" * 32, dict(st, steps=0))] * 2, "frame")      # a device that rendered nothing is not a right answer
    assert not idle["valid"]
    flaky = bench.validation_block([("ab" * 32, dict(st))] * 2, "frame", repeat_digest="ee" * 32)
    assert not flaky["valid"] and not flaky["repeat_identical_on_device0"]
    json.dumps(good)
    assert bench.forms_valid(None) and bench.forms_valid({"batch": {"frames_identical": True}, "rgb8_batch": {"frames_identical": None}, "png_files": {"skipped": "x"}})
    assert not bench.forms_valid({"batch": {"frames_identical": False}}) and not bench.forms_valid({"split": {"identical_to_one_device": False}})
    # frames as numpy arrays or torch tensors hash by their bytes
    a = np.arange(24, dtype=np.float64).reshape(2, 4, 3)
    import torch
    assert bench.frame_digest(np, a) == bench.frame_digest(np, torch.from_numpy(a.copy())) != bench.frame_digest(np, a + 1)
    assert bench.digest_as_float("f" * 64) == float(0xFFFFFFFFFFFF)


def test_d2h_forms_compare_delivered_frames_when_the_frames_are_the_same_scene():
    """same_frames: after the timed call every delivered frame must equal the first -- on this rank and (48 bits of a digest through the
    float collective) on every other; one differing frame, or one rank with another frame, makes frames_identical False."""
    state = {"poison": None}

    class Bs:
        @staticmethod
        def alloc_image(tree, h, w, dtype=np.float64):
            return np.zeros((h, w, 3), dtype)

        @staticmethod
        def render_batch(cfgs, trees, outs=None):
            for i, o in enumerate(outs):
                o[:] = 7
            if state["poison"] is not None:
                outs[state["poison"]][0, 0, 0] = 8
            return outs

    common = (Bs, np, ["t0", "t1"], ["same"] * 12, 16, 8, 1, ["batch"], lambda: None, lambda x: x)
    assert bench.d2h_forms(*common)["batch"]["frames_identical"] is None                      # frames differ by design (animation): not compared
    assert bench.d2h_forms(*common, same_frames=True)["batch"]["frames_identical"] is True
    state["poison"] = 5
    assert bench.d2h_forms(*common, same_frames=True)["batch"]["frames_identical"] is False
    state["poison"] = None
    seen = []
    r = bench.d2h_forms(*common, same_frames=True, all_ranks=lambda x: (seen.append(x), [x, x])[1])
    assert r["batch"]["frames_identical"] is True and len(seen) == 1 and seen[0] >= 0
    r = bench.d2h_forms(*common, same_frames=True, all_ranks=lambda x: [x, x + 1.0])          # the other rank delivered another frame
    assert r["batch"]["frames_identical"] is False
    state["poison"] = 0
    seen.clear()
    r = bench.d2h_forms(*common, same_frames=True, all_ranks=lambda x: (seen.append(x), [x, 5.0])[1])
    assert r["batch"]["frames_identical"] is False and seen == [-1.0]                        # still a collective: every rank takes part
    # the split leg is supplied by the caller
    r = bench.d2h_forms(*common[:7], ["split"], *common[8:], split=lambda: {"Mpixel_s": 1.0, "identical_to_one_device": True})
    assert r == {"split": {"Mpixel_s": 1.0, "identical_to_one_device": True}}


def test_split_leg_single_process_and_per_rank():
    """--form split: bs_render_split over N contexts in one process, or band `rank` of world per process; both compared with ONE device's
    whole frame byte for byte, with the one-device time of the same blocking call beside it."""
    H, W = 2160, 3840
    calls = []

    def pixel(rows):   # a frame whose content depends on the absolute row only
        return np.broadcast_to(np.arange(rows.start, rows.stop, dtype=np.float64)[:, None, None], (len(rows), W, 3))

    class Tree:
        def stats(self):
            return {"rays": 4 * H * W, "steps": 123456789, "wave_iters": 1000, "kernel_ms": 19.0}

    class Bs:
        Config = None

        @staticmethod
        def alloc_image(tree, h, w, dtype=np.float64):
            return np.zeros((h, w, 3), dtype)

        @staticmethod
        def render(cfg, tree, out=None):
            calls.append("render")
            out[:] = pixel(range(0, H))
            return out

        @staticmethod
        def render_split(cfg, trees, out=None):
            calls.append(("split", len(trees)))
            out[:] = pixel(range(0, H))
            return out

        @staticmethod
        def render_rows(cfg, tree, row0, row1, out=None):
            calls.append(("rows", row0, row1))
            out[:] = pixel(range(row0, row1))
            return out

    import blackstar_amd as real
    Bs.Config = real.Config
    r = bench.split_leg(Bs, np, [Tree(), Tree(), Tree()], 0, 1, lambda: None, lambda x: x, lambda o: [o], reps=2)
    assert r["identical_to_one_device"] and r["parts"] == 3 and r["entry_point"] == "bs_render_split" and r["bands"] == [[0, 720], [720, 1440], [1440, 2160]]
    assert ("split", 3) in calls and r["one_device_steps"] == 123456789 and abs(r["speedup_vs_one_device"] - r["one_device_ms_per_frame"] / r["ms_per_frame"]) < 1e-9
    assert r["frames"] == 1 and abs(r["Mpixel_s"] - W * H / r["seconds"] / 1e6) < 1e-6
    # ... and what an 8-GPU split will be limited by, from one device: the 8 bands rendered alone, their steps and times, the bounds
    p = r["prediction_8_gpus"]
    assert p["n_bands"] == 8 and p["bands"][0] == [0, 270] and p["bands"][-1] == [1890, 2160] and len(p["band_steps"]) == len(p["band_kernel_ms"]) == len(p["band_call_ms"]) == 8
    assert [c for c in calls if c[0] == "rows"][:2] == [("rows", 0, 270), ("rows", 270, 540)] and p["steps_max_over_mean"] == 1.0 and p["work_bound"] == 8.0
    assert len([c for c in calls if c[0] == "rows"]) == 8 * 4        # one untimed pass over the bands, then three of which the fastest counts
    assert p["predicted_speedup_bound"] > 0 and abs(p["kernel_bound"] - 1.0) < 1e-12 and abs(p["fixed_ms_per_band"] - (8 * 19.0 - 19.0) / 8) < 1e-9
    calls.clear()
    # rank 1 of 4: its band against its own whole frame, and every rank's whole frame is the same frame
    gathered = []

    def gather(o):
        gathered.append(o)
        return [(True, o[1], (0, 540)), o, (True, o[1], (1080, 1620)), (True, o[1], (1620, 2160))]

    r = bench.split_leg(Bs, np, [Tree()], 1, 4, lambda: None, lambda x: x, gather, reps=1)
    assert ("rows", 540, 1080) in calls and r["identical_to_one_device"] and r["parts"] == 4 and r["bands"][1] == [540, 1080]
    r = bench.split_leg(Bs, np, [Tree()], 1, 4, lambda: None, lambda x: x, lambda o: [(True, "other frame", (0, 540)), o], reps=1)
    assert not r["identical_to_one_device"]
    # the headline built from it: strong scaling, one frame
    args = types.SimpleNamespace(steps=20, warmup=3, mode="fast", traffic_bytes=None, catalogue="synthetic", workload="default-aa")
    st = {"steps": 1854063332, "rays": 8294400, "wave_iters": 29008800, "kernel_ms": 4.4}
    res = bench.result_line(args, 4, "x", r["Mpixel_s"], r["seconds"], W, H, None, st, 1.0)
    bench.split_headline(args, res, r, 4)
    assert res["scaling"] == "strong" and res["steps"] == 1 and res["config"]["baseline_config"] == "configs[3]" and "split" in res and "roofline" in res
    assert res["config"]["parallelism"] == "row bands x4" and "lensing-disk" in res["metric"]
    json.dumps(res)


def test_parity_block_counts_what_is_outside_the_bar():
    """SURVEY.md 8d 'Parity check': per config, the count of values outside |gpu - cpu| <= 1e-4 |cpu| + 1e-7, max abs / rel error, and
    whether the step and fate counters equal the oracle's -- in the driver-run line, not only in builder-run reports."""
    rng = np.random.default_rng(5)
    ref = rng.random((9, 16, 3)) * 2
    st = {"rays": 144, "steps": 32000, "capped": 0, "horizon": 10, "escaped": 134, "disk_hits": 20, "star_hits": 30}
    same = bench.parity_block(np, "cfg", ref, st, ref.copy(), dict(st), "strict")
    assert same["outside_1e-4"] == 0 and same["bit_identical"] and same["steps_equal"] and same["fates_equal"] and same["max_abs"] == 0 and same["values"] == ref.size
    near = bench.parity_block(np, "cfg", ref, st, ref * (1 + 5e-5), dict(st), "fast")
    assert near["outside_1e-4"] == 0 and not near["bit_identical"] and 4e-5 < near["max_rel_where_ref>1e-3"] < 6e-5
    got = ref.copy()
    got[3, 5, 1] *= 1 + 3e-4
    got[0, 0, 0] = np.nan
    off = bench.parity_block(np, "cfg", ref, st, got, dict(st, steps=st["steps"] - 1, horizon=11), "fast")
    assert off["outside_1e-4"] == 2 and off["nonfinite"] == 1 and not off["steps_equal"] and not off["fates_equal"]
    assert {(p["y"], p["x"], p["channel"]) for p in off["first_outside"]} == {(3, 5, 1), (0, 0, 0)}
    assert off["counters_oracle_gpu"]["steps"] == [32000, 31999]
    assert bench.parity_block(np, "cfg", ref, st, ref[:4], dict(st), "fast")["outside_1e-4"] == ref.size
    json.dumps(off)


def test_cpu_baseline_carries_parity_for_all_five_baseline_configs(monkeypatch):
    """cpu_baseline keeps the oracle's frames (it used to throw them away) and compares them with what the product renders of the same
    configs: FAST on the timed workload's sample (configs[2]), FAST + STRICT on configs[1], FAST on configs[0], and -- round 6 -- FAST on
    configs[3] (lensing-disk, down-scaled, supersampled) and on frames 0 / 300 / 599 of configs[4] (the animation).  One value outside the
    bar, or a step count that differs, makes parity_ok False (and bench.py then prints "valid": false)."""
    from oracle import c_oracle, scenes
    from blackstar_amd import synthetic
    monkeypatch.setattr(scenes, "DEFAULT", scenes.with_res(scenes.DEFAULT, 160, 90))    # (the whole 1080p frames take the oracle seconds each)
    monkeypatch.setattr(bench, "PARITY_C4_RES", (64, 36))
    monkeypatch.setattr(bench, "PARITY_C5_RES", (48, 27))
    star_bytes = synthetic.catalogue_bytes("synthetic")
    ix = c_oracle.Index(c_oracle.read_ppm(star_bytes))
    calls = []

    def product(c, with_stars, mode):   # stands in for bs_render here (no GPU): the same arithmetic as the checker
        calls.append((c["width"], c["height"], with_stars, mode))
        return c_oracle.render(c, ix if with_stars else c_oracle.Index(None), threads=0)

    cfg = dict(scenes.DEFAULT_AA)
    blk = bench.cpu_baseline(cfg, star_bytes, 0.05, product, np)
    assert blk["kind"] == "port" and blk["cores"] >= 1 and blk["value"] > 0
    assert [(p["mode"]) for p in blk["parity"]] == ["fast", "fast", "strict", "fast", "fast", "fast", "fast", "fast"] and blk["parity_ok"]
    assert all(p["outside_1e-4"] == 0 and p["bit_identical"] and p["steps_equal"] and p["fates_equal"] for p in blk["parity"])
    assert [p["baseline_config"] for p in blk["parity"]] == ["configs[2]", "configs[1]", "configs[1]", "configs[0]", "configs[3]", "configs[4]", "configs[4]", "configs[4]"]
    assert blk["parity_configs"] == ["configs[0]", "configs[1]", "configs[2]", "configs[3]", "configs[4]"]
    assert "lensing-disk" in blk["parity"][4]["config"] and "frame 300" in blk["parity"][6]["config"] and "frame 599" in blk["parity"][7]["config"]
    assert calls[0][2] is True and calls[1][2] is False and calls[3][:2] == (640, 480) and "1e-4" in blk["parity_tolerance"]
    assert calls[4] == (64, 36, True, "fast") and calls[7] == (48, 27, True, "fast")
    assert blk["parity"][4]["counters_oracle_gpu"]["rays"] == [4 * 64 * 36] * 2            # supersampled like the config itself

    def wrong(c, with_stars, mode):
        img, st = product(c, with_stars, mode)
        img.flat[int(np.argmax(img))] *= 1.001      # (one value of each frame, 10x outside the bar)
        return img, st

    assert not bench.cpu_baseline(cfg, star_bytes, 0.05, wrong, np)["parity_ok"]

    def crashes(c, with_stars, mode):
        raise RuntimeError("bs_render: BS_EDEVICE")

    blk = bench.cpu_baseline(cfg, star_bytes, 0.05, crashes, np)
    assert not blk["parity_ok"] and all("BS_EDEVICE" in p["error"] for p in blk["parity"])
    assert "parity" not in bench.cpu_baseline(cfg, star_bytes, 0.05)      # no product to check (nothing renders on the CPU instead)
    # A rank that has bound itself to its GPU's NUMA node (bind_rank_to_gpu_node) still measures the CPU baseline on ALL the cores the
    # process had before: host_cpus; the binding comes back afterwards.
    if hasattr(os, "sched_setaffinity") and len(os.sched_getaffinity(0)) >= 2:
        everything = os.sched_getaffinity(0)
        one = {min(everything)}
        os.sched_setaffinity(0, one)
        try:
            assert bench.cpu_baseline(cfg, star_bytes, 0.05)["cores"] == 1                      # bound: one core is all it sees
            wide = bench.cpu_baseline(cfg, star_bytes, 0.05, host_cpus=everything)
            assert wide["cores"] > 1 and os.sched_getaffinity(0) == one                          # the host's cores, and the binding is back
        finally:
            os.sched_setaffinity(0, everything)


def test_launcher_torchrun_with_one_gpu_forces_the_distributed_branch(monkeypatch):
    """`bench.py --gpus 1 --launcher torchrun` = one rank under torch.distributed.run with BLACKSTAR_BENCH_FORCE_DIST=1: the whole RCCL
    branch of run_ranks (init_process_group, all_gathers, barrier, all_gather_object, gather, destroy) executes on a 1-GPU box."""
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--launcher", "torchrun", "--gather"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert seen["env"]["BLACKSTAR_BENCH_FORCE_DIST"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "--nproc-per-node=1" in seen["cmd"] and "127.0.0.1" in seen["cmd"] and seen["cmd"][-5:] == ["--gpus", "1", "--launcher", "torchrun", "--gather"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'dist_on = world > 1 or os.environ.get("BLACKSTAR_BENCH_FORCE_DIST") == "1"' in src
    assert "if world > 1:\n        dist." not in src            # every collective of run_ranks hangs on dist_on, so the forced run executes them all
