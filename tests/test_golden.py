"""The committed fixtures of tests/golden/ (made by tests/golden/make_golden.py from the CPU oracle).

CPU part: the oracle still produces them bit for bit (it cannot drift unnoticed).
GPU part: the HIP path, through the C ABI, produces the same accumulator words, counters and per-bounce
hit / material ids WITHOUT the oracle being run — the fixtures alone are the reference.
The reference repository holds no golden vectors of its own; these pin the restatement, not the reference.
"""
import hashlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H, SPP = 64, 36, 16
DEPTHS = {"c1": 4, "c2": 8, "c3": 10, "c4": 11, "c5": 16, "cpu_tracer": 11, "colored_mesh": 6, "box_sides": 6}
CONFIGS = ("c1", "c2", "c3", "c4", "c5", "cpu_tracer", "colored_mesh", "box_sides")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def scene(rta, name):
    return rta.scenes.CONFIGS[name](width=W, height=H, spp=SPP, max_depth=DEPTHS[name])


def test_rng_and_sampler_vectors(orc):
    g = load("rng_and_samplers.npz")
    for (s, p, k), want_key, want in zip(g["keys"], g["path_keys"], g["draws"]):
        key = orc.lib.orc_path_key(int(s), int(p), int(k))
        assert key == int(want_key)
        got = np.array([orc.lib.orc_draw(key, i) for i in range(16)], dtype=np.float32)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    us = g["u"]
    sc = np.array([orc.sincos2pi(float(u)) for u in us], dtype=np.float32)
    np.testing.assert_array_equal(sc.view(np.uint32), g["sincos"].view(np.uint32))
    uv = np.array([orc.unit_vector(float(us[i]), float(us[(7 * i + 3) % 64])) for i in range(64)], dtype=np.float32)
    np.testing.assert_array_equal(uv.view(np.uint32), g["unit_vector"].view(np.uint32))
    he = np.array([orc.hemisphere_glsl((0.6, 0.0, 0.8), float(us[i]), float(us[(5 * i + 1) % 64])) for i in range(64)], dtype=np.float32)
    np.testing.assert_array_equal(he.view(np.uint32), g["hemisphere"].view(np.uint32))


def test_shader_random_vectors(orc):
    import ctypes as C
    rows = load("shader_random.npz")["rows"]
    args = ((1, 64, 36, 0, 0, 0), (1000, 1920, 1080, 1919, 1079, 99), (77, 1080, 720, 500, 360, 250))
    for want, a in zip(rows, args):
        st = (C.c_float * 3)()
        orc.lib.orc_glsl_seed(*a, st)
        got = np.array(list(st) + [orc.lib.orc_glsl_random(st) for _ in range(13)], dtype=np.float32)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_host_side_fixtures(rta, orc):
    g = load("host_side.npz")
    cams = [(1920, 1080, (1, 1, 6)), (1080, 720, (1, 1, 6)), (512, 512, (1, 0, 6)), (3840, 2160, (1, 0, 6))]
    for want, (w, h, pos) in zip(g["cameras"], cams):
        np.testing.assert_array_equal(rta.Camera(w, h, 1.0, pos).data.view(np.uint32), want.view(np.uint32))
    for f, tag in (("Bunny-LowPoly.off", "low"), ("bunny.off", "big")):
        tris = rta.scenes.main_cpp_mesh(f)                     # through librt_mi355x.so's host code
        assert len(tris) == int(g[tag + "_n"][0])
        assert hashlib.sha256(tris.tobytes()).digest() == g[tag + "_sha"].tobytes()
        assert tris[:4].tobytes() == g[tag + "_first4"].tobytes()


@pytest.mark.parametrize("name", CONFIGS)
def test_oracle_reproduces_scene_fixture(rta, orc, name):
    g = load(f"scene_{name}.npz")
    cfg = scene(rta, name)
    assert cfg.shapes.tobytes() == g["shapes"].tobytes() and len(cfg.triangles) == int(g["n_triangles"][0])
    np.testing.assert_array_equal(cfg.camera.view(np.uint32), g["camera"].view(np.uint32))
    p = orc.params(W, H, SPP, cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode,
                   flags=cfg.flags, triangle_albedo=cfg.triangle_albedo, triangle_colors=cfg.triangle_colors)
    acc, st = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    np.testing.assert_array_equal(acc, g["accum"])
    assert [st[k] for k in ("samples", "segments", "diffuse", "specular", "terminal")] == list(g["counters"])
    hit, mat = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, p, 9)
    np.testing.assert_array_equal(hit, g["hit_s9"]); np.testing.assert_array_equal(mat, g["mat_s9"])
    img = orc.resolve(acc, SPP, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))
    np.testing.assert_array_equal(img.view(np.uint32), g["image"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CONFIGS)
def test_hip_path_reproduces_scene_fixture(rta, name):
    g = load(f"scene_{name}.npz")
    cfg = scene(rta, name)
    rt = rta.scenes.make_renderer(cfg)
    try:
        grid_n = 4 if cfg.sample_mode != rta.SAMPLE_JITTER else 0          # 16 samples = renderPixel<4>'s grid
        st = rt.render(W, H, SPP, cfg.max_depth, cfg.seed, grid_n=grid_n, stats=True)
        img = rt.resolve(W, H, gamma_sqrt=(cfg.semantics == 0))
        np.testing.assert_array_equal(img.view(np.uint32), g["image"].view(np.uint32))
        assert [st[k] for k in ("samples", "segments", "diffuse", "specular", "terminal")] == list(g["counters"])
        for s, tag in ((0, "s0"), (9, "s9")):
            hit, mat = rt.traceDebug(W, H, cfg.max_depth, cfg.seed, s, grid_n=grid_n)
            np.testing.assert_array_equal(hit, g["hit_" + tag]); np.testing.assert_array_equal(mat, g["mat_" + tag])
    finally:
        rt.close()
