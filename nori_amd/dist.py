"""One-process-per-GPU sharding of a render (torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" for the CPU tests).

The path shards with no data-path collective: every rank renders its share of
the frame into its own zero-initialised RGBW buffer; the only exchange is one
SUM-reduce of that buffer to rank 0 -- ImageBlock::put(ImageBlock&)
(src/block.cpp:93-102) across GPUs instead of across TBB workers.

  mode "tile"   : 16x16 tiles round-robin over ranks (tile_mod / tile_rem);
                  fixed total work -> strong scaling.
  mode "sample" : every rank renders the whole frame with a disjoint range of
                  the per-pixel sample indices (disjoint pcg32 streams).
Both give, up to float summation order, the single-GPU image.
"""
from __future__ import annotations


def shard(mode: str, rank: int, world: int, spp: int):
    """kwargs for render_into / render_host of rank `rank`."""
    if mode == "tile":
        return dict(spp_begin=0, spp_count=spp, tile_mod=world, tile_rem=rank)
    if mode == "sample":
        base, extra = divmod(spp, world)
        begin = rank * base + min(rank, extra)
        return dict(spp_begin=begin, spp_count=base + (1 if rank < extra else 0), tile_mod=1, tile_rem=0)
    raise ValueError(f"unknown shard mode {mode!r}")


def reduce_frame(frame, dst: int = 0):
    """SUM-reduce the RGBW frame tensor to rank `dst` (in place there)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


def render_distributed(render_fn, frame, mode: str, spp: int, rank: int, world: int, **kw):
    """render_fn(frame, **shard_kwargs, **kw) accumulates this rank's share into
    `frame` (a torch tensor, CUDA for the product / CPU in tests); then reduce."""
    frame.zero_()
    stats = render_fn(frame, **shard(mode, rank, world, spp), **kw)
    reduce_frame(frame, 0)
    return stats
