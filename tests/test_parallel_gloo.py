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
