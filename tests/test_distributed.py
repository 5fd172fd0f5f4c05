"""world_size-2 gloo test (CPU) of the frame-sharded multi-process path: sharding map + final gather order."""
import os
import socket
import sys

import pytest

from blackstar_amd.distributed import owner_of, shard_frames


def test_shard_map():
    assert shard_frames(7, 0, 2) == [0, 2, 4, 6] and shard_frames(7, 1, 2) == [1, 3, 5]
    assert shard_frames(600, 3, 8) == list(range(3, 600, 8))
    assert sorted(sum((shard_frames(13, r, 4) for r in range(4)), [])) == list(range(13))
    assert shard_frames(0, 0, 2) == [] and owner_of(11, 8) == 3
    with pytest.raises(ValueError):
        shard_frames(4, 2, 2)


def test_more_ranks_than_frames_fails_on_every_rank_before_any_collective():
    """world > n_frames with a gather: every rank raises up front (no rank may enter dist.gather while another fails)."""
    from blackstar_amd.distributed import render_sharded

    class NoCollectives:  # any collective call would be a hang in production
        def gather(self, *a, **k):
            raise AssertionError("entered a collective")

    calls = []
    for rank in range(3):
        with pytest.raises(ValueError):
            render_sharded(2, lambda i: calls.append(i), rank, 3, gather_to=0, dist=NoCollectives())
    assert calls == []
    # without a gather an idle rank is fine
    assert render_sharded(2, lambda i: i, 2, 3, gather_to=None, dist=NoCollectives()) == []


def test_no_frames_at_all_is_an_empty_result_not_an_error():
    """n_frames == 0 (an empty directory in app/Main.hs:68-77's batch loop): no rounds, no collective; [] on the root, None elsewhere."""
    from blackstar_amd.distributed import render_sharded

    def never(i):
        raise AssertionError("no frame to render")
    assert render_sharded(0, never, 0, 4, gather_to=0) == []
    assert render_sharded(0, never, 3, 4, gather_to=0) is None
    assert render_sharded(0, never, 1, 4, gather_to=None) == []


def _worker(rank, world, port, n_frames, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blackstar_amd.distributed import render_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rendered = []

    def fake_render(i):  # stands in for bs.render_device into a torch tensor: content identifies frame and rank
        rendered.append(i)
        return torch.full((4, 6, 3), float(i) + 0.001 * rank, dtype=torch.float64)

    frames = render_sharded(n_frames, fake_render, rank, world, gather_to=0)
    dist.barrier()
    q.put((rank, rendered, None if frames is None else [float(f[0, 0, 0]) for f in frames]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [4, 5])
def test_two_process_frame_sharding_and_gather(n_frames):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, rendered, frames = q.get(timeout=120)
        res[r] = (rendered, frames)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == list(range(0, n_frames, 2)) and res[1][0] == list(range(1, n_frames, 2))
    assert res[1][1] is None
    assert res[0][1] == [i + 0.001 * (i % 2) for i in range(n_frames)]  # frame order restored on the root, each from its owner


def test_row_shard_map():
    from blackstar_amd.distributed import shard_rows
    assert [shard_rows(1080, r, 8) for r in range(8)] == [(135 * r, 135 * (r + 1)) for r in range(8)]
    bands = [shard_rows(2161, r, 8) for r in range(8)]  # ragged: sizes differ by at most one, contiguous, cover everything
    assert bands[0][0] == 0 and bands[-1][1] == 2161 and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
    assert {b - a for a, b in bands} == {270, 271}
    assert shard_rows(3, 2, 4) == (2, 3) and shard_rows(3, 3, 4) == (3, 3)
    with pytest.raises(ValueError):
        shard_rows(10, 4, 4)


def _row_worker(rank, world, port, height, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blackstar_amd.distributed import render_frame_by_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bands = []

    def fake_band(row0, row1):  # stands in for bs.render_rows_device: every row carries its own index, every band its rank
        bands.append((row0, row1))
        t = torch.arange(row0, row1, dtype=torch.float64).reshape(-1, 1, 1).expand(row1 - row0, 5, 3).clone()
        t[:, 0, 0] += 0.001 * rank
        return t

    frame = render_frame_by_rows(height, 5, fake_band, rank, world, gather_to=0)
    dist.barrier()
    q.put((rank, bands, None if frame is None else (tuple(frame.shape), [float(v) for v in frame[:, 1, 2]], [float(v) for v in frame[:, 0, 0]])))
    dist.destroy_process_group()


@pytest.mark.parametrize("height", [8, 9])
def test_two_process_row_sharding_of_one_frame(height):
    import torch.multiprocessing as mp
    from blackstar_amd.distributed import shard_rows
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_row_worker, args=(r, 2, port, height, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, bands, frame = q.get(timeout=120)
        res[r] = (bands, frame)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == [shard_rows(height, 0, 2)] and res[1][0] == [shard_rows(height, 1, 2)]
    assert res[1][1] is None
    shape, rows, tagged = res[0][1]
    assert shape == (height, 5, 3) and rows == [float(y) for y in range(height)]  # bands re-assembled in row order, padding trimmed
    split = shard_rows(height, 0, 2)[1]
    assert tagged == [y + (0.001 if y >= split else 0.0) for y in range(height)]  # each row from the rank that owns it


def _bench_validation_worker(rank, world, port, poison_rank, q, crash=False):
    """bench.py's untimed validation as two real processes over gloo: the objects all-gather of the validation block, the float collective
    of the delivered forms' frames_identical, the per-rank split leg.  poison_rank renders a frame that differs in one value."""
    import numpy as np
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import blackstar_amd as real
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 2160, 3840
    bad = rank == poison_rank

    def frame(r0, r1):
        a = np.broadcast_to(np.arange(r0, r1, dtype=np.float64)[:, None, None], (r1 - r0, W, 3)).copy()
        if bad:
            a[-1, -1, -1] += 1.0
        return a

    class Tree:
        def stats(self):
            return {"rays": 4 * H * W, "steps": 1000 + (1 if bad else 0), "wave_iters": 10, "kernel_ms": 19.0}

    class Bs:
        Config = real.Config

        @staticmethod
        def alloc_image(tree, h, w, dtype=np.float64):
            return np.zeros((h, w, 3), dtype)

        @staticmethod
        def render(cfg, tree, out=None):
            out[:] = frame(0, H)
            return out

        @staticmethod
        def render_rows(cfg, tree, row0, row1, out=None):
            if bad and crash:
                raise RuntimeError("bs_render_rows: BS_EDEVICE (test)")
            out[:] = frame(row0, row1)
            return out

        @staticmethod
        def render_batch(cfgs, trees, outs=None):
            if bad and crash:
                raise RuntimeError("bs_render_batch: BS_ENOMEM (test)")
            for o in outs:
                o[:] = 7 + (1 if bad else 0)
            return outs

    def all_ranks(x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        got = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        return [float(g.item()) for g in got]

    def gather_objs(o):
        objs = [None] * world
        dist.all_gather_object(objs, o)
        return objs

    split = bench.split_leg(Bs, np, [Tree()], rank, world, dist.barrier, lambda x: max(all_ranks(x)), gather_objs, reps=1)
    forms = bench.d2h_forms(Bs, np, [Tree()], ["same"] * 6, 16, 8, world, ["batch"], dist.barrier, lambda x: max(all_ranks(x)), same_frames=True, all_ranks=all_ranks)
    digest = bench.frame_digest(np, frame(0, 4))
    val = bench.validation_block(gather_objs((digest, {k: (1000 + (1 if bad else 0) if k == "steps" else 5) for k in bench.COUNTERS})), "frame")
    dist.barrier()
    if crash:
        q.put((rank, split.get("error"), forms["batch"].get("error"), bench.forms_valid({"split": split, "batch": forms["batch"]}),
               bench.legs_failed({"split": split, "batch": forms["batch"]})))
    else:
        q.put((rank, split["identical_to_one_device"], split["bands"], split["parts"], forms["batch"]["frames_identical"], val["valid"],
               val["frames_identical_across_devices"], val["steps_per_device"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("poison_rank", [-1, 1])
def test_two_process_bench_validation_tells_a_wrong_frame_on_one_rank(poison_rank):
    """VERDICT r3 item 1b as real processes: with both ranks right every check says so on every rank; with rank 1 wrong in ONE value of its
    frames every rank -- rank 0 too, which prints the line -- learns it: split not identical, delivered frames not identical, validation
    invalid with the per-rank step counters.  No rank hangs in a collective the other one skipped."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_validation_worker, args=(r, 2, port, poison_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        got = q.get(timeout=180)
        res[got[0]] = got[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ok = poison_rank < 0
    for r in (0, 1):
        split_ok, bands, parts, forms_ok, valid, frames_same, steps = res[r]
        assert bands == [[0, 1080], [1080, 2160]] and parts == 2
        assert split_ok is ok and forms_ok is ok and valid is ok and frames_same is ok, (r, res[r])
        assert steps == ([1000, 1000] if ok else [1000, 1001])


def test_two_process_bench_legs_survive_a_rank_whose_render_fails():
    """The N > 1 line must not be lost -- or hang -- because an OPTIONAL leg fails on one rank: rank 1's bs_render_rows / bs_render_batch
    raise; both ranks still pass every fence and collective, both report {"error": ...} for the split leg and the delivered form (rank 1 its
    own message, rank 0 "another rank failed").  forms_valid only speaks about frames that were compared; the failed legs are NAMED
    (legs_failed -> `legs_failed` in the line, which bench.py prints with "valid": false: a product entry point failed on a device)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_validation_worker, args=(r, 2, port, 1, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        got = q.get(timeout=180)
        res[got[0]] = got[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "BS_EDEVICE" in res[1][0] and "BS_ENOMEM" in res[1][1]
    assert res[0][0] == "another rank failed" and res[0][1] == "another rank failed"
    assert res[0][2] is True and res[1][2] is True
    assert res[0][3] == ["batch", "split"] and res[1][3] == ["batch", "split"]
