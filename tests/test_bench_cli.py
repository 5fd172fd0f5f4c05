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
    assert a.sustained_frames == 500 and a.cpu_seconds > 0


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


def test_d2h_forms_drive_the_products_batch_entry_points_with_a_ring_of_four_buffers_per_context():
    """with_d2h: frame i goes to context i % N through ONE bs_render_batch / bs_render_rgb8_batch call; two frames are in flight per
    context, so a ring of 4 page-locked buffers per context is enough; the warm-up touches every ring buffer once."""
    calls = []

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
        def alloc_png(tree, h, w):
            return np.zeros(h * w * 3 + 100, np.uint8)

        @staticmethod
        def render_png_batch(cfgs, trees, outs=None):
            calls.append(("png", len(cfgs), len(trees), [id(o) for o in outs], outs[0].dtype))
            return [memoryview(o)[:40 + i % 2] for i, o in enumerate(outs)]   # files of 40 and 41 bytes

    trees = ["t0", "t1", "t2"]
    frames = [f"cfg{i}" for i in range(30)]
    fences = []
    res = bench.d2h_forms(FakeBs, np, trees, frames, 16, 8, 1, ["batch", "rgb8-batch", "png-batch", "png-files"], lambda: fences.append(1), lambda x: x)
    assert set(res) == {"batch", "rgb8_batch", "png_batch", "png_files"} and len(fences) == 8
    assert [c[:3] for c in calls] == [("batch", 30, 3), ("batch", 30, 3), ("rgb8", 30, 3), ("rgb8", 30, 3), ("png", 30, 3), ("png", 30, 3),
                                      ("files", 30, 3), ("files", 30, 3)]  # warm-up call, timed call
    assert res["png_files"]["bytes_written_per_frame"] == 77 and res["png_files"]["frames"] == 30 and res["png_files"]["entry_point"] == "bs_render_png_files"
    del calls[6:]
    assert res["png_batch"]["entry_point"] == "bs_render_png_batch" and res["png_batch"]["bytes_to_host_per_frame"] == 40   # the mean file size
    host = res["png_batch"]["host_encoder_baseline"]      # the same frame through zlib on one host core, beside the device encoder's number
    assert set(host) == {"zlib_level1", "zlib_level6"} and all(v["bytes"] > 0 and v["ms_per_frame_one_core"] >= 0 for v in host.values())
    assert calls[1][4] == np.float64 and calls[3][4] == np.uint8
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
    assert "disk full" in r["error"] and len(fences) == 2 and collectives == [0.0, float("inf")]   # same fences and collectives as a healthy rank


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
