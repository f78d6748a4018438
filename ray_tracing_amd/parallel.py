"""Splitting one frame over the GPUs of a node, one process per GPU.

The reference is single-GPU (one GL context); this has no counterpart there.
Every sample of every pixel is independent and its random stream depends only
on (seed, pixel, sample), so a frame can be cut two ways:

* rows    — rank r renders the rows y with (y // band) % world == r.  The
            accumulators of the ranks are disjoint, so their element-wise sum
            is bit-identical to a single-GPU render.
* samples — rank r renders samples [r*spp, (r+1)*spp) of every pixel; the sum
            of the accumulators is a frame of world*spp samples per pixel.

Both end in the same exchange: ONE sum-reduce of the accumulator to rank 0
(`torch.distributed.reduce`, RCCL on GPUs, gloo on CPU), followed by the
resolve on rank 0.  With float accumulators (CPU-tracer semantics) the sample
split changes the order of the float additions, so it matches a single-GPU
render to rounding, not bitwise; with the shader semantics' integer
accumulators it is exact either way.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist

from ._abi import SEM_CPU


@dataclass
class Shard:
    mode: str            # "rows" | "samples"
    rank: int
    world: int
    band_rows: int       # rows
    first_sample: int    # samples
    spp: int             # samples this rank traces per pixel


def plan(mode, rank, world, spp, band_rows=16):
    """What rank `rank` of `world` renders when every rank is given `spp` samples per pixel."""
    if mode == "rows":
        return Shard("rows", rank, world, band_rows, 0, spp)
    if mode == "samples":
        return Shard("samples", rank, world, band_rows, rank * spp, spp)
    raise ValueError(f"unknown shard mode {mode!r}")


def apply(renderer, shard):
    """Configure a RayTracingComputeShader for its shard."""
    if shard.mode == "rows":
        renderer.setRowShard(shard.rank, shard.world, shard.band_rows)
    else:
        renderer.setRowShard(0, 1, shard.band_rows)


def total_samples(shard):
    """Samples per pixel in the reduced frame."""
    return shard.spp if shard.mode == "rows" else shard.spp * shard.world


def accum_view(accum_words, semantics):
    """The accumulator (int32 words, [h, w, 4]) as the dtype its sums are kept in."""
    return accum_words.view(torch.float32) if semantics == SEM_CPU else accum_words


def reduce_framebuffer(accum_words, semantics, dst=0, group=None, async_op=False):
    """Sum the ranks' accumulators into rank `dst` (in place there).

    With async_op the collective is only queued (behind the work already on the current stream) and
    its handle returned: the caller can launch the next frame's trace before calling .wait(), so
    that the exchange over xGMI runs beside it."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    t = accum_view(accum_words, semantics)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no reduce for device tensors; its all_reduce gives rank `dst` the same sum (rehearsals only)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
