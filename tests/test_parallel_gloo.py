"""The N > 1 path on CPU: two gloo ranks shard a frame the way bench.py's GPU ranks do
(parallel.plan / parallel.exchange_framebuffer), with the CPU oracle standing in for the kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, config, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel
    from oracle import oracle as orc
    cfg = rta.scenes.CONFIGS[config](width=48, height=40, spp=6, max_depth=5)
    shard = parallel.plan(mode, rank, world, cfg.spp, band_rows=16)
    p = orc.params(cfg.width, cfg.height, shard.spp, cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics,
                   first_sample=shard.first_sample, triangle_albedo=cfg.triangle_albedo,
                   shard=(shard.rank, shard.world, shard.band_rows) if mode == "rows" else (0, 1, 16), threads=2)
    acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    t = torch.from_numpy(acc.view(np.int32).copy())
    parallel.exchange_framebuffer(t, cfg.semantics, shard)     # rows: a gather of the owned bands; samples: one reduce(SUM) to rank 0
    if rank == 0:
        np.save(out_path, t.numpy().view(np.uint32))
        assert parallel.total_samples(shard) == (cfg.spp if mode == "rows" else cfg.spp * world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,config", [("rows", "c1"), ("samples", "c1"), ("samples", "c3"), ("rows", "c3")])
def test_two_ranks_reassemble_the_frame(rta, orc, tmp_path, mode, config):
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, _free_port(), mode, config, out), nprocs=2, join=True)
    got = np.load(out)
    cfg = rta.scenes.CONFIGS[config](width=48, height=40, spp=6, max_depth=5)
    total = cfg.spp if mode == "rows" else 2 * cfg.spp
    p = orc.params(cfg.width, cfg.height, total, cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics,
                   triangle_albedo=cfg.triangle_albedo)
    want, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    if mode == "rows" or cfg.semantics == rta.SEM_GLSL:
        # disjoint rows, or integer sums: bit-identical to one rank
        np.testing.assert_array_equal(got, want)
    else:
        # float sums split by sample range: same samples, different order of additions
        np.testing.assert_allclose(got.view(np.float32), want.view(np.float32), rtol=2e-6, atol=1e-6)
        assert (got != want).mean() < 0.5


def test_plan_covers_every_row_and_sample():
    from ray_tracing_amd import parallel
    for world in (1, 2, 4, 8):
        rows = [set(y for y in range(1080) if (y // 16) % world == r) for r in range(world)]
        assert set().union(*rows) == set(range(1080)) and sum(len(r) for r in rows) == 1080
        shards = [parallel.plan("samples", r, world, 100) for r in range(world)]
        covered = sorted(s for sh in shards for s in range(sh.first_sample, sh.first_sample + sh.spp))
        assert covered == list(range(100 * world))


def _gather_worker(rank, world, port, height, band, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ray_tracing_amd import parallel
    shard = parallel.plan("rows", rank, world, 1, band_rows=band)
    # every rank's frame holds its rank + 1 in its own rows and a poison value elsewhere
    t = torch.full((height, 5, 4), -7, dtype=torch.int32)
    t[parallel.owned_rows(height, rank, world, band)] = rank + 1
    handle = parallel.gather_bands(t, shard, async_op=True)
    handle.wait()
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,band", [(2, 40, 16), (3, 50, 8), (2, 7, 16)])
def test_band_gather_puts_every_rank_s_rows_in_place(tmp_path, world, height, band):
    """Ragged cases: a last band that is cut short, ranks with different row counts, a rank without any row."""
    out = str(tmp_path / "frame.npy")
    mp.spawn(_gather_worker, args=(world, _free_port(), height, band, out), nprocs=world, join=True)
    got = np.load(out)
    want = ((np.arange(height) // band) % world + 1).astype(np.int32)
    np.testing.assert_array_equal(got, np.broadcast_to(want[:, None, None], got.shape))


def test_exchange_bytes():
    from ray_tracing_amd import parallel
    frame = 1920 * 1080 * 16
    assert parallel.exchange_bytes("rows", 1920, 1080, 8) == frame // 8
    assert parallel.exchange_bytes("samples", 1920, 1080, 8) == frame * 7 // 8
    assert parallel.exchange_bytes("rows", 1920, 1080, 1) == 0


def _steady_gather_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ray_tracing_amd import parallel
    height, band = 70, 16
    shard = parallel.plan("rows", rank, world, 1, band_rows=band)
    # two accumulators used in turn, as bench.py's frames use them
    frames = [torch.zeros((height, 3, 4), dtype=torch.int32) for _ in range(2)]
    made, ptrs, allocs = [], [], []
    real_empty_like, real_arange = torch.empty_like, torch.arange
    counted = {"n": 0}

    def counting(fn):
        def wrapped(*a, **k):
            counted["n"] += 1
            return fn(*a, **k)
        return wrapped
    for k in range(6):
        t = frames[k % 2]
        t.fill_(-7)
        t[parallel.owned_rows(height, rank, world, band)] = 100 * k + rank + 1
        if k >= 2:                                              # from the third frame on both accumulators have their buffers
            torch.empty_like, torch.arange = counting(real_empty_like), counting(real_arange)
        h = parallel.gather_bands(t, shard, async_op=True)
        h.wait()
        torch.empty_like, torch.arange = real_empty_like, real_arange
        made.append(parallel.band_buffer_builds)
        ptrs.append(tuple(b.send.data_ptr() for b in parallel._band_buffers.values()))
        if rank == 0:
            allocs.append(t[:, 0, 0].clone().numpy())
    assert made == [1, 2, 2, 2, 2, 2], made                    # one set of buffers per accumulator, then none
    assert ptrs[1] == ptrs[5] and len(ptrs[5]) == 2            # ... and the same memory every frame
    assert counted["n"] == 0, counted                           # no index list rebuilt, no receive buffer allocated
    if rank == 0:
        np.save(out_path, np.stack(allocs))
    dist.barrier()
    dist.destroy_process_group()


def test_band_gather_allocates_nothing_after_its_first_frame(tmp_path):
    out = str(tmp_path / "frames.npy")
    mp.spawn(_steady_gather_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)                                         # [frame, row]
    for k in range(6):
        np.testing.assert_array_equal(got[k], 100 * k + (np.arange(70) // 16) % 2 + 1)
