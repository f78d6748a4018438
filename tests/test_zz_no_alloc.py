"""rt_render_noalloc — the render call that allocates nothing — against rt_render, through the C ABI.

Same samples into caller-owned memory, bit for bit; two calls in a row on one GPU, the second with a smaller
frame (the static scratch still holds the first call's costs and plan); sizes beyond the scratch are refused.
Collected last (the file name sorts after the other tests): a failure here must not hide them.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes():
    """Free device memory as the HIP runtime reports it (no allocation involved)."""
    import torch
    return torch.cuda.mem_get_info()[0]


@pytest.mark.parametrize("name", ["c3", "c4"])
def test_noalloc_matches_rt_render_two_calls_in_a_row(rta, name):
    import torch
    from ray_tracing_amd import _abi
    lib = _abi.lib
    # first a frame whose plan cuts tiles into slices (c3, shader sums) or orders them (c4, float sums); then a smaller one
    sizes = [dict(width=323, height=181, spp=96, max_depth=6), dict(width=131, height=77, spp=70, max_depth=6)]
    cfgs = [rta.scenes.CONFIGS[name](**kw) for kw in sizes]
    want = []
    for cfg in cfgs:                                   # the existing entry point, a fresh context each
        rt = rta.scenes.make_renderer(cfg)
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
        rt.render(cfg.width, cfg.height, 5, cfg.max_depth, cfg.seed, first_sample=cfg.spp)     # a short call: no probe
        bound = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
        rt.bindAccum(bound.data_ptr(), cfg.width, cfg.height)
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
        rt.render(cfg.width, cfg.height, 5, cfg.max_depth, cfg.seed, first_sample=cfg.spp)
        torch.cuda.synchronize()
        want.append(bound.cpu().numpy().view(np.uint32))
        rt.bindAccum(None, 0, 0)
        rt.close()

    rt = rta.scenes.make_renderer(cfgs[0])             # one context, one scene, both frame sizes
    outs = [torch.zeros(c.height, c.width, 4, dtype=torch.int32, device="cuda") for c in cfgs]
    torch.cuda.synchronize()
    free0 = _free_bytes()
    for cfg, out in zip(cfgs, outs):
        cam = rta.Camera(cfg.width, cfg.height)
        cam.data[:] = cfg.camera
        rt.camera = cam
        rt.updateCameraBuffer()
        for first, n in ((0, cfg.spp), (cfg.spp, 5)):
            rc = lib.rt_render_noalloc(rt._ctx, C.c_void_p(out.data_ptr()), cfg.width, cfg.height, n, cfg.max_depth, cfg.seed, first, 0)
            assert rc == 0, lib.rt_last_error(rt._ctx).decode()
    torch.cuda.synchronize()
    assert _free_bytes() == free0                      # nothing was allocated on the device
    for cfg, out, ref in zip(cfgs, outs, want):
        got = out.cpu().numpy().view(np.uint32)
        diff = got != ref
        assert not diff.any(), f"{cfg.width}x{cfg.height}: {int(diff.sum())} of {diff.size} words differ from rt_render's"
        assert ref[..., :3].any()
    rt.close()


def test_noalloc_limits(rta):
    import torch
    from ray_tracing_amd import _abi
    lib = _abi.lib
    cfg = rta.scenes.config1(width=64, height=64, spp=4, max_depth=4)
    rt = rta.scenes.make_renderer(cfg)
    out = torch.zeros(64, 64, 4, dtype=torch.int32, device="cuda")
    # more tiles than the static scratch holds: refused before anything is launched
    w = 16 * 1024
    h = 16 * (_abi.NOALLOC_MAX_TILES // 1024) + 16
    assert lib.rt_render_noalloc(rt._ctx, C.c_void_p(out.data_ptr()), w, h, 1, 2, 1, 0, 0) == -1
    assert b"static scratch" in lib.rt_last_error(rt._ctx)
    assert lib.rt_render_noalloc(rt._ctx, None, 64, 64, 1, 2, 1, 0, 0) == -1
    assert lib.rt_render_noalloc(rt._ctx, C.c_void_p(out.data_ptr() + 4), 64, 64, 1, 2, 1, 0, 0) == -1
    # a scene changed and not yet built: the build allocates, so this entry point will not do it
    rt.shapes = cfg.shapes[:2].copy()
    s = rt.shapes
    assert lib.rt_upload_shapes(rt._ctx, s.ctypes.data, len(s)) == 0
    assert lib.rt_render_noalloc(rt._ctx, C.c_void_p(out.data_ptr()), 64, 64, 1, 2, 1, 0, 0) == -5
    assert lib.rt_commit_scene(rt._ctx) == 0
    assert lib.rt_render_noalloc(rt._ctx, C.c_void_p(out.data_ptr()), 64, 64, 1, 2, 1, 0, 0) == 0
    torch.cuda.synchronize()
    assert not out[..., 3].any()
    rt.close()
