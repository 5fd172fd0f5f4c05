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
