#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ from the CPU oracle (oracle/rt_oracle.c).

The reference holds no golden vectors and cannot be built or seeded here, so these are outputs of the
oracle itself: they pin it (and, through tests/test_golden.py, the HIP path) against drift.  Run from
the repository root after an intended change of the oracle's arithmetic:

    python tests/golden/make_golden.py

Writes small .npz files (numpy arrays only, no pickles).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import ray_tracing_amd as rta          # host-side constructors only (no GPU needed)
from oracle import oracle as orc

W, H, SPP = 64, 36, 16
DEPTHS = {"c1": 4, "c2": 8, "c3": 10, "c4": 11, "c5": 16, "cpu_tracer": 11, "colored_mesh": 6, "box_sides": 6}


def scene(name):
    # "cpu_tracer" samples on renderPixel<N>'s grid: 16 = 4 x 4
    return rta.scenes.CONFIGS[name](width=W, height=H, spp=SPP, max_depth=DEPTHS[name])


def main():
    # (i) RNG stream vectors: first 16 draws of four (seed, pixel, sample) keys
    keys = [(1, 0, 0), (1, 77, 3), (0xdeadbeefcafe, 2073599, 99), (7, 12345, 255)]
    draws = np.array([[orc.lib.orc_draw(orc.lib.orc_path_key(s, p, k), i) for i in range(16)] for s, p, k in keys], dtype=np.float32)
    pkeys = np.array([orc.lib.orc_path_key(s, p, k) for s, p, k in keys], dtype=np.uint32)
    hashes = np.array([orc.lib.orc_hash32(x) for x in (0, 1, 2, 0xdeadbeef, 0xffffffff, 12345)], dtype=np.uint32)
    us = (np.arange(64, dtype=np.float32) + np.float32(0.25)) / np.float32(64)
    sincos = np.array([orc.sincos2pi(float(u)) for u in us], dtype=np.float32)
    unit = np.array([orc.unit_vector(float(us[i]), float(us[(7 * i + 3) % 64])) for i in range(64)], dtype=np.float32)
    hemi = np.array([orc.hemisphere_glsl((0.6, 0.0, 0.8), float(us[i]), float(us[(5 * i + 1) % 64])) for i in range(64)], dtype=np.float32)
    import ctypes as C
    glsl = []
    for seed, w, h, gx, gy, smp in ((1, 64, 36, 0, 0, 0), (1000, 1920, 1080, 1919, 1079, 99), (77, 1080, 720, 500, 360, 250)):
        st = (C.c_float * 3)()
        orc.lib.orc_glsl_seed(seed, w, h, gx, gy, smp, st)
        glsl.append(list(st) + [orc.lib.orc_glsl_random(st) for _ in range(13)])
    np.savez_compressed(os.path.join(HERE, "shader_random.npz"), rows=np.array(glsl, dtype=np.float32))
    np.savez_compressed(os.path.join(HERE, "rng_and_samplers.npz"), keys=np.array(keys, dtype=np.uint64), path_keys=pkeys,
                        draws=draws, hashes=hashes, u=us, sincos=sincos, unit_vector=unit, hemisphere=hemi)

    # (ii)+(iii) per configuration scene at 64x36, 16 spp: accumulator words, counters, per-bounce hit / material ids
    for name in ("c1", "c2", "c3", "c4", "c5", "cpu_tracer", "colored_mesh", "box_sides"):
        cfg = scene(name)
        p = orc.params(W, H, SPP, cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode,
                       flags=cfg.flags, triangle_albedo=cfg.triangle_albedo, triangle_colors=cfg.triangle_colors)
        acc, st = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
        hit0, mat0 = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, p, 0)
        hit9, mat9 = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, p, 9)
        img = orc.resolve(acc, SPP, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))
        np.savez_compressed(os.path.join(HERE, f"scene_{name}.npz"), accum=acc, image=img,
                            counters=np.array([st[k] for k in ("samples", "segments", "diffuse", "specular", "terminal")], dtype=np.uint64),
                            hit_s0=hit0.astype(np.int16), mat_s0=mat0.astype(np.uint32), hit_s9=hit9.astype(np.int16), mat_s9=mat9.astype(np.uint32),
                            camera=cfg.camera, shapes=cfg.shapes.view(np.uint8).reshape(len(cfg.shapes), -1),
                            n_triangles=np.array([len(cfg.triangles)]))

    # (iv) host-side constructors: camera blocks, the mesh pipeline of main.cpp:108-122
    cams = np.array([orc.camera_init(w, h, 1.0, pos) for w, h, pos in ((1920, 1080, (1, 1, 6)), (1080, 720, (1, 1, 6)), (512, 512, (1, 0, 6)), (3840, 2160, (1, 0, 6)))])
    mesh = {}
    for f in ("Bunny-LowPoly.off", "bunny.off"):
        tris, nv = orc.mesh_triangles(os.path.join(rta.scenes.MESH_DIR, f))
        mesh[f] = (len(tris), nv, hashlib.sha256(tris.tobytes()).hexdigest(), tris[:4].copy().view(np.uint8))
    np.savez_compressed(os.path.join(HERE, "host_side.npz"), cameras=cams,
                        low_n=np.array(mesh["Bunny-LowPoly.off"][:2]), low_sha=np.frombuffer(bytes.fromhex(mesh["Bunny-LowPoly.off"][2]), dtype=np.uint8),
                        low_first4=mesh["Bunny-LowPoly.off"][3],
                        big_n=np.array(mesh["bunny.off"][:2]), big_sha=np.frombuffer(bytes.fromhex(mesh["bunny.off"][2]), dtype=np.uint8),
                        big_first4=mesh["bunny.off"][3])
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
