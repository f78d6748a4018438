"""Splitting one frame over the GPUs of a node, one process per GPU.

The reference is single-GPU (one GL context); this has no counterpart there.
Every sample of every pixel is independent and its random stream depends only
on (seed, pixel, sample), so a frame can be cut two ways:

* rows    — rank r renders the rows y with (y // band) % world == r.  The
            accumulators of the ranks are disjoint, so the assembled frame is
            bit-identical to a single-GPU render.  The exchange is a GATHER
            of each rank's own bands to rank 0: width*height*16/world bytes
            per rank on the wire.
* samples — rank r renders samples [r*spp, (r+1)*spp) of every pixel; the sum
            of the accumulators is a frame of world*spp samples per pixel.
            The exchange is ONE sum-reduce of the accumulator to rank 0
            (width*height*16 bytes through every rank of the ring).

Either way the exchange is a single collective (`torch.distributed`, RCCL on
GPUs, gloo on CPU) followed by the resolve on rank 0.  With float accumulators
(CPU-tracer semantics) the sample split changes the order of the float
additions, so it matches a single-GPU render to rounding, not bitwise; with
the shader semantics' integer accumulators it is exact either way.
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
    """Samples per pixel in the exchanged frame."""
    return shard.spp if shard.mode == "rows" else shard.spp * shard.world


def accum_view(accum_words, semantics):
    """The accumulator (int32 words, [h, w, 4]) as the dtype its sums are kept in."""
    return accum_words.view(torch.float32) if semantics == SEM_CPU else accum_words


def owned_rows(height, rank, world, band_rows):
    """Image rows of rank `rank`: y with (y // band_rows) % world == rank (rt_set_row_shard's rule), ascending."""
    y = torch.arange(height)
    return y[(y // band_rows) % world == rank]


def exchange_bytes(mode, width, height, world):
    """Bytes each rank puts on the wire for one frame (for predictions; the frame has 16 bytes per pixel)."""
    frame = width * height * 16
    if world <= 1:
        return 0
    # ring reduce: every rank forwards (world - 1) / world of the frame; band gather: a rank sends its own bands once
    return frame * (world - 1) // world if mode == "samples" else frame // world


def reduce_framebuffer(accum_words, semantics, dst=0, group=None, async_op=False, always=False):
    """Sum the ranks' accumulators into rank `dst` (in place there): the exchange of the `samples` split.

    With async_op the collective is only queued (behind the work already on the current stream) and
    its handle returned: the caller can launch the next frame's trace before calling .wait(), so
    that the exchange over xGMI runs beside it.  A single-rank group returns None without a
    collective unless `always` is set (tests run the RCCL call itself that way)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always):
        return None
    t = accum_view(accum_words, semantics)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # Rehearsals only (ranks sharing one GPU, gloo between them): summed through host memory and completed
        # here, whatever async_op says (the call blocks on the copy to the host) — gloo's own handling of device
        # tensors was seen to stall on the second 33 MB frame.
        host = t.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) == dst:
            t.copy_(host)
        return None
    return dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class _BandGather:
    """Handle of a queued band gather: .wait() completes it and, on the destination, puts the bands in place."""

    def __init__(self, work, place):
        self._work, self._place = work, place

    def wait(self):
        if self._work is not None:
            self._work.wait()
        if self._place is not None:
            self._place()
            self._place = None
        return True


class _BandBuffers:
    """What one accumulator's band gathers need besides the accumulator itself, made once and kept: the row index list
    of every rank (on the accumulator's device), the send buffer, and on the destination the receive buffers."""

    def __init__(self, accum_words, world, rank, band_rows, dst, on_host):
        height = accum_words.shape[0]
        dev = accum_words.device
        self.rows = [owned_rows(height, r, world, band_rows).to(dev) for r in range(world)]
        most = max(len(r) for r in self.rows)
        self.send = accum_words.new_zeros((most,) + tuple(accum_words.shape[1:]))
        # rehearsals (ranks sharing one GPU, gloo between them): gloo gathers host tensors only
        self.send_host = torch.zeros_like(self.send, device="cpu") if on_host else None
        wire = self.send_host if on_host else self.send
        self.recv = [torch.empty_like(wire) for _ in range(world)] if rank == dst else None
        self.back = [torch.empty_like(self.send) for _ in range(world)] if (on_host and rank == dst) else None


_band_buffers = {}
band_buffer_builds = 0          # how often a _BandBuffers was made (tests: a second frame makes none)


def _buffers_for(accum_words, world, rank, band_rows, dst, on_host, group):
    global band_buffer_builds
    key = (accum_words.data_ptr(), tuple(accum_words.shape), accum_words.dtype, str(accum_words.device), world, rank, band_rows, dst,
           on_host, id(group))
    b = _band_buffers.get(key)
    if b is None:
        if len(_band_buffers) >= 16:                     # accumulators come and go (bench.py measures two splits)
            _band_buffers.clear()
        b = _band_buffers[key] = _BandBuffers(accum_words, world, rank, band_rows, dst, on_host)
        band_buffer_builds += 1
    return b


def gather_bands(accum_words, shard, dst=0, group=None, async_op=False, always=False):
    """The exchange of the `rows` split: every rank sends the bands it rendered to rank `dst`, which puts them into
    its own accumulator (in place).  A rank's bands are copied into one contiguous send buffer (padded to the
    largest rank's row count, so that all contributions have one size), one gather moves them, and `dst` writes
    rank r's rows back with the same row index list.  Nothing is summed: the frame is bit-identical to a single-GPU
    render whatever the sums' type.

    The index lists and the send / receive buffers belong to the accumulator (keyed by its address) and are made by
    its first gather only: a frame's exchange is one index_select into the send buffer, the gather, and on `dst`
    world - 1 index_copy_ — no allocation, no host-to-device copy.  Two accumulators used in turn have two sets, so
    a gather still in flight is not overwritten by the next frame's.  With gloo on device tensors (rehearsals) the
    bands go through host buffers and the call completes before it returns, whatever async_op says."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always):
        return None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_host = accum_words.is_cuda and dist.get_backend(group) == "gloo"
    b = _buffers_for(accum_words, world, rank, shard.band_rows, dst, on_host, group)
    n_own = len(b.rows[rank])
    if n_own:
        torch.index_select(accum_words, 0, b.rows[rank], out=b.send[:n_own])
    if on_host:
        b.send_host.copy_(b.send)
        dist.gather(b.send_host, b.recv, dst=dst, group=group)
        if rank == dst:
            for r in range(world):
                if r != dst and len(b.rows[r]):
                    b.back[r].copy_(b.recv[r])
                    accum_words.index_copy_(0, b.rows[r], b.back[r][:len(b.rows[r])])
        return _BandGather(None, None) if async_op else None
    work = dist.gather(b.send, b.recv, dst=dst, group=group, async_op=async_op)

    def place():
        for r in range(world):
            if r != dst and len(b.rows[r]):
                accum_words.index_copy_(0, b.rows[r], b.recv[r][:len(b.rows[r])])

    handle = _BandGather(work if async_op else None, place if rank == dst else None)
    if async_op:
        return handle
    handle.wait()
    return None


def exchange_framebuffer(accum_words, semantics, shard, dst=0, group=None, async_op=False, always=False):
    """The one exchange of a frame: a band gather for the `rows` split, a sum-reduce for the `samples` split."""
    if shard.mode == "rows":
        return gather_bands(accum_words, shard, dst=dst, group=group, async_op=async_op, always=always)
    return reduce_framebuffer(accum_words, semantics, dst=dst, group=group, async_op=async_op, always=always)
