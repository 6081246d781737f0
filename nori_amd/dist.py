"""One-process-per-GPU sharding of a render (torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" for the CPU tests).  Started by `bench.py --gpus N`
(which spawns the ranks itself) or by any torch.distributed launcher.

The path shards with no data-path collective: every rank renders its share of
the frame into its own zero-initialised RGBW buffer; the only exchange is ONE
merge of those buffers on rank 0 -- ImageBlock::put(ImageBlock&)
(src/block.cpp:93-102) across GPUs instead of across TBB workers.

  split "tile"   : 16x16 tiles round-robin over ranks (tile_mod / tile_rem);
                   fixed total work -> strong scaling.
  split "sample" : every rank renders the whole frame with a disjoint range of
                   the per-pixel sample indices (disjoint pcg32 streams).
  merge "reduce" : SUM-reduce of the whole RGBW frame to rank 0 (either split).
  merge "gather" : tile split only.  With tiles_x divisible by the world size a rank's
                   tiles are whole tile COLUMNS (c = rank mod world), so what it
                   touched is a set of (16 + 2 border)-pixel-wide column strips: each rank
                   packs its strips, one gather brings them to rank 0, which adds them
                   (overlapping halos included) into the frame -- 1/N of the frame per
                   rank over the wire instead of a ring reduce of all of it.
Both give, up to float summation order, the single-GPU image.

film_order = "reference" (the reference's own summation order, bit for bit): render_distributed_reference -- every rank
renders whole ROWS of 32x32 blocks into an array of block accumulators, ONE reduce of those (disjoint, hence exact) arrays,
and rank 0 adds the blocks into the frame in BlockGenerator's order: the frame has the bits of the one-GPU frame for any
number of ranks.
"""
from __future__ import annotations

TILE = 16      # NORI_TILE_SIZE of the device code (rt_types.h kTile)


def shard(mode: str, rank: int, world: int, spp: int):
    """kwargs for render_into / render_host of rank `rank`."""
    if mode == "tile":
        return dict(spp_begin=0, spp_count=spp, tile_mod=world, tile_rem=rank)
    if mode == "sample":
        base, extra = divmod(spp, world)
        begin = rank * base + min(rank, extra)
        return dict(spp_begin=begin, spp_count=base + (1 if rank < extra else 0), tile_mod=1, tile_rem=0)
    raise ValueError(f"unknown shard mode {mode!r}")


def block_rows(rank: int, world: int, n_rows: int):
    """(row_begin, row_count) of rank `rank`: contiguous rows of 32x32 blocks, the first n_rows % world ranks one more
    (group_merge.h group_block_rows)."""
    base, extra = divmod(n_rows, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def _group_up() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def reduce_frame(frame, dst: int = 0):
    """SUM-reduce the RGBW frame tensor to rank `dst` (in place there).  Runs whenever a process group exists --
    also for a group of one rank, so that a single-GPU launch under torch.distributed exercises the RCCL call."""
    import torch.distributed as dist
    if _group_up():
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


def column_strips(rank: int, world: int, tiles_x: int, border: int, frame_cols: int, device=None):
    """Bordered-frame x coordinates rank `rank`'s tile columns touch (tile column c covers
    [16 c, 16 c + 16 + 2 border)), and which of them to pack: those inside the frame (the last column of an image
    whose width is not a multiple of 16 is clipped) at their FIRST occurrence -- a rank's own strips overlap when
    16 world < 16 + 2 border (always for a group of one rank), and a frame column packed twice would be added twice."""
    import torch
    cols = torch.arange(rank, tiles_x, world, device=device)
    x = (cols[:, None] * TILE + torch.arange(TILE + 2 * border, device=device)[None, :]).reshape(-1)
    first = torch.ones_like(x, dtype=torch.bool)
    if x.numel() > 1:
        first[1:] = x[1:] > torch.cummax(x, 0).values[:-1]
    return x.clamp(max=frame_cols - 1), (x < frame_cols) & first


def gather_frame(frame, rank: int, world: int, tiles_x: int, border: int, dst: int = 0):
    """Tile-split merge by ONE gather of every rank's column strips (see module docstring)."""
    import torch
    import torch.distributed as dist
    if tiles_x % world != 0:
        raise ValueError(f"gather merge needs tiles_x ({tiles_x}) divisible by the world size ({world}); use merge='reduce'")
    if not _group_up():
        return frame
    x, valid = column_strips(rank, world, tiles_x, border, frame.shape[1], frame.device)
    pack = (frame[:, x, :] * valid[None, :, None].to(frame.dtype)).contiguous()
    parts = [torch.empty_like(pack) for _ in range(world)] if rank == dst else None
    dist.gather(pack, parts, dst=dst)
    if rank == dst:
        frame.zero_()
        for r in range(world):
            xr, _ = column_strips(r, world, tiles_x, border, frame.shape[1], frame.device)
            frame.index_add_(1, xr, parts[r])      # masked entries are zero: clamped and repeated columns add nothing
    return frame


def render_distributed(render_fn, frame, mode: str, spp: int, rank: int, world: int, merge: str = "reduce",
                       tiles_x: int | None = None, border: int = 0, merge_ms: list | None = None, **kw):
    """render_fn(frame, **shard_kwargs, **kw) accumulates this rank's share into
    `frame` (a torch tensor, CUDA for the product / CPU in tests); then merge on rank 0.
    merge_ms: a list that receives this rank's time in the exchange (events on the frame's stream; it contains the wait for
    the slowest rank) -- only when a process group exists."""
    import time
    frame.zero_()
    stats = render_fn(frame, **shard(mode, rank, world, spp), **kw)
    timed = merge_ms is not None and _group_up()
    if timed and frame.is_cuda:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    t0 = time.perf_counter()
    if merge == "gather":
        if mode != "tile":
            raise ValueError("gather merge applies to the tile split only")
        gather_frame(frame, rank, world, tiles_x, border, 0)
    else:
        reduce_frame(frame, 0)
    if timed and frame.is_cuda:
        e1.record(); e1.synchronize()
        merge_ms.append(float(e0.elapsed_time(e1)))
    elif timed:
        merge_ms.append((time.perf_counter() - t0) * 1e3)
    return stats


def render_distributed_reference(renderer, frame, spp: int, rank: int, world: int, merge_ms: list | None = None, **kw):
    """film_order = reference over the ranks (module docstring).  `renderer`: nori_amd.render.Renderer with
    set_option("film_order", "reference") (anything with its block_rows / block_acc_floats / render_block_rows_into /
    resolve_blocks: the CPU tests pass a stand-in); `frame`: this rank's RGBW tensor, the merged frame on rank 0."""
    import contextlib
    import time
    import torch
    # A caller's stream (kw["stream"], a torch.cuda.Stream) carries EVERYTHING of this call: the zeroing of the accumulators,
    # the render, the reduce (torch.distributed orders a collective after the current stream's work), the frame's zeroing and the
    # block merge -- each step reads what the previous one wrote, and a non-blocking stream is ordered against no other.
    stream = kw.pop("stream", None)
    on = torch.cuda.stream(stream) if (stream is not None and frame.is_cuda) else contextlib.nullcontext()
    with on:
        acc = torch.zeros(renderer.block_acc_floats(), dtype=torch.float32, device=frame.device)
        r0, rn = block_rows(rank, world, renderer.block_rows())
        if stream is not None:
            kw["stream"] = stream
        stats = renderer.render_block_rows_into(acc, r0, rn, spp_count=spp, **kw)
        timed = merge_ms is not None and _group_up()
        if timed and frame.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = time.perf_counter()
        reduce_frame(acc, 0)      # disjoint arrays: x + 0 = x, exact whatever order the ring adds in
        frame.zero_()
        if rank == 0:
            if stream is not None:
                renderer.resolve_blocks(acc, frame, stream=stream)
            else:
                renderer.resolve_blocks(acc, frame)
        if timed and frame.is_cuda:
            e1.record(); e1.synchronize()
            merge_ms.append(float(e0.elapsed_time(e1)))
        elif timed:
            merge_ms.append((time.perf_counter() - t0) * 1e3)
    return stats
