"""Frame sharding for animation batches: one process per GPU, frame i -> rank i % world.

The reference renders a directory of frames sequentially in one process (app/Main.hs:68-77).  Frames are
independent, so the multi-GPU path has NO data-path collective: every rank renders its own frames with its
own bs_ctx; the only communication is the optional gather of finished frames to rank 0 (RCCL over xGMI when the
tensors live in HBM, gloo on CPU in tests).  A single huge frame can instead be split into row bands, one per GPU
(shard_rows / render_frame_by_rows over bs_render_rows).  torch.distributed is plumbing here, nothing more.
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


def shard_rows(height: int, rank: int, world: int) -> "tuple[int, int]":
    """The band of output rows rank `rank` renders when ONE frame is split over `world` GPUs: contiguous, sizes differ by at
    most one row, empty (row0 == row1) only if there are more ranks than rows.  Every ray is independent and a supersampled
    output row only needs its own two traced rows, so bands need no halo (the fallback for single huge frames)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(max(height, 0), world)
    row0 = rank * base + min(rank, extra)
    return row0, row0 + base + (1 if rank < extra else 0)


def render_frame_by_rows(height: int, width: int, render_band: Callable[[int, int], "object"], rank: int, world: int,
                         gather_to: Optional[int] = 0, dist=None):
    """One frame, row-sharded: this rank renders `render_band(row0, row1) -> (row1-row0, width, 3) tensor`; the bands are
    gathered on `gather_to` (padded to a common height for the collective, trimmed after) and returned there as the
    (height, width, 3) frame; other ranks get None.  With gather_to=None the local band is returned as it is."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    if world > height:  # checked identically on every rank, before anything is rendered or any collective is entered
        raise ValueError(f"{world} ranks for a frame of {height} rows: use world <= height")
    row0, row1 = shard_rows(height, rank, world)
    band = render_band(row0, row1)
    if gather_to is None or world == 1:
        return band
    rows_max = -(-height // world)
    padded = torch.zeros((rows_max, width, 3), dtype=band.dtype, device=band.device)
    padded[: row1 - row0] = band
    if rank == gather_to:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, bufs, dst=gather_to)
        parts = []
        for k in range(world):
            a, b = shard_rows(height, k, world)
            parts.append(bufs[k][: b - a])
        return torch.cat(parts, dim=0)
    dist.gather(padded, None, dst=gather_to)
    return None


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
    if gather_to is not None and world > 1 and 0 < n_frames < world:  # (no frames at all: no rounds, [] on the root)
        # checked identically on every rank BEFORE anything is rendered or any collective is entered: a rank without a single
        # frame has no tensor shape to contribute to the gather, and failing there alone would leave the other ranks hanging
        raise ValueError(f"{world} ranks for {n_frames} frames: use world <= n_frames (or gather_to=None)")
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
            t = torch.zeros_like(template)  # template is set: world <= n_frames gives every rank a frame in round 0
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


def _png_bound(height, width):
    from .raytracer import png_bound
    return png_bound(height, width)


def _frame_namer(n_frames: int, basename: str, reference_names: bool):
    """<basename>_<index>.png: zero-padded to the width of the last index, or -- reference_names -- exactly as `animate` + the batch loop name
    the frame (app/Animate.hs:55-56 with src/Util.hs:43-48 padZero, whose index 0 comes out unpadded: blackstar_amd.animation.pad_zero)."""
    from .animation import frame_file_name
    width = len(str(max(n_frames - 1, 1)))
    if reference_names:
        return lambda j: frame_file_name(basename, n_frames, j, ".png")
    return lambda j: f"{basename}_{j:0{width}d}.png"


def write_animation(animation, tree, out_dir: str, rank: int = 0, world: int = 1, basename: str = "frame", pipe: int = 16,
                    reference_names: bool = False) -> "list[str]":
    """app/Animate.hs + blackstar's batch mode (app/Main.hs:68-77) for one animation: frame i on rank i % world, each rendered, bloomed,
    mapped to sRGB8, ENCODED AS A PNG FILE on the device and written by the library (`bs_render_png_files`: frames in flight on the GPU,
    a native writer thread writing `<basename>_<zero-padded index>.png` from page-locked buffers meanwhile).  No collective, no pixels on
    the host, no Python between frames.  `pipe`: the ring of file buffers per context.  reference_names: the reference's own file names
    (padZero's quirk for index 0 included) instead of plain zero padding.  Returns the paths this rank wrote."""
    import os

    from .animation import generate_frames, validate_keyframes
    from .batch import render_png_files

    validate_keyframes(animation.keyframes)
    frames = generate_frames(animation)
    name = _frame_namer(len(frames), basename, reference_names)
    os.makedirs(out_dir, exist_ok=True)
    mine = shard_frames(len(frames), rank, world)
    paths = [os.path.join(out_dir, name(j)) for j in mine]
    render_png_files([frames[j] for j in mine], [tree], paths, pipe=pipe)
    return paths


def render_animation(animation, tree, rank: int = 0, world: int = 1, gather_to: Optional[int] = 0, out_dir: Optional[str] = None,
                     basename: str = "frame", dist=None, reference_names: bool = False):
    """BASELINE configs[4] end to end: the frames of an Animation (src/Animation.hs generateFrames), frame i on rank
    i % world, each through the device pipeline of doRender (render -> bloom -> sRGB8, `bs_render_rgb8_batch`), gathered as
    RGB8 on `gather_to` (6.2 MB per 1080p frame instead of 49.8 MB of f64).  With out_dir, every rank also writes the frames it
    rendered as PNG files (`<basename>_<zero-padded index>.png`, the naming of app/Animate.hs:55-56 with the padding done
    right -- SURVEY Appendix F.7; reference_names=True: exactly the reference's names), encoded on its GPU (`bs_encode_png`); write_animation is the variant that ONLY writes files
    and never brings pixels to the host.  Returns the ordered list of (h, w, 3) uint8 tensors on the root, None elsewhere."""
    import os

    import torch

    from .animation import generate_frames, validate_keyframes
    from .batch import render_rgb8_batch
    from .raytracer import alloc_png, encode_png

    validate_keyframes(animation.keyframes)
    frames = generate_frames(animation)
    name = _frame_namer(len(frames), basename, reference_names)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)

    # This rank's frames go through the device pipeline kPipe at a time (`bs_render_rgb8_batch`: two frames in flight, the next
    # frame's trace kernel overlaps this frame's tail and bloom); render_sharded then asks for them in order.
    kPipe = 16
    mine = shard_frames(len(frames), rank, world)
    ready = {}
    # The files are encoded on the GPU (well under a millisecond per 1080p frame, against 0.1-0.25 s of zlib on a host core) right here
    # -- a context is driven by one thread at a time -- and written by a worker thread while the GPU goes on.
    pool = None
    pending = []
    png_buf = None
    if out_dir:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=2)

    def write_file(path, data):
        with open(path, "wb") as f:
            f.write(data)

    def one(i):
        if i not in ready:
            pos = mine.index(i)
            chunk = mine[pos:pos + kPipe]
            for j, img in zip(chunk, render_rgb8_batch([frames[j] for j in chunk], [tree])):
                ready[j] = img
        rgb8 = ready.pop(i)
        if pool is not None:
            nonlocal png_buf
            if png_buf is None or png_buf.size < _png_bound(*rgb8.shape[:2]):
                png_buf = alloc_png(tree, *rgb8.shape[:2])
            pending.append(pool.submit(write_file, os.path.join(out_dir, name(i)), bytes(encode_png(rgb8, tree, out=png_buf))))
        return torch.from_numpy(rgb8)

    try:
        return render_sharded(len(frames), one, rank, world, gather_to=gather_to, dist=dist)
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
            for f in pending:
                f.result()  # a failed write raises here
