"""Frame sharding for animation batches: one process per GPU, frame i -> rank i % world.

The reference renders a directory of frames sequentially in one process (app/Main.hs:68-77).  Frames are
independent, so the multi-GPU path has NO data-path collective: every rank renders its own frames with its
own bs_ctx; the only communication is the final gather of finished frames to rank 0 (RCCL over xGMI when the
tensors live in HBM, gloo on CPU in tests).  torch.distributed is plumbing here, nothing more.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence


def shard_frames(n_frames: int, rank: int, world: int) -> List[int]:
    """Indices of the frames rank `rank` renders: round-robin, so neighbouring (similar-cost) frames spread evenly."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(range(rank, max(n_frames, 0), world))


def owner_of(frame: int, world: int) -> int:
    return frame % world


def render_sharded(n_frames: int, render_frame: Callable[[int], "object"], rank: int, world: int, gather_to: Optional[int] = 0,
                   dist=None) -> Optional[Sequence]:
    """Render this rank's frames with `render_frame(i) -> tensor` and gather them, in frame order, on `gather_to`.

    Rounds of `world` frames: in round r rank k renders frame r*world + k; after each round one gather moves
    that round's frames to the root (each peer over its own xGMI link).  Ranks without a frame in the last,
    ragged round contribute a zero tensor that the root drops.  Returns the ordered frame list on the root,
    None elsewhere (or the local frames if gather_to is None)."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    mine = shard_frames(n_frames, rank, world)
    out: List = []
    rounds = (n_frames + world - 1) // world
    template = None
    for r in range(rounds):
        i = r * world + rank
        t = render_frame(i) if i < n_frames else None
        if t is not None:
            template = t
        if gather_to is None:
            if t is not None:
                out.append(t)
            continue
        if t is None:
            if template is None:
                raise RuntimeError("rank has no frame at all; use world <= n_frames")
            t = torch.zeros_like(template)
        if world == 1:
            out.append(t)
            continue
        if rank == gather_to:
            bufs = [torch.empty_like(t) for _ in range(world)]
            dist.gather(t, bufs, dst=gather_to)
            for k in range(world):
                if r * world + k < n_frames:
                    out.append(bufs[k])
        else:
            dist.gather(t, None, dst=gather_to)
    assert gather_to is not None or len(out) == len(mine)
    if gather_to is None or rank == gather_to:
        return out
    return None
