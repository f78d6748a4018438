"""Host-side constructors (camera, shapes, SimpleTriangle, SimpleMesh) against the reference's constants
and against the oracle's independent restatement."""
import math
import os

import numpy as np
import pytest


def test_camera_constants_1080p(rta):
    # camera.h:14-28 at 1920x1080: unit-area image plane, worldStep = 1/1440
    cam = rta.Camera(1920, 1080)
    assert cam.getWorldStep() == np.float32(np.float32(1920 / math.sqrt(1920 * 1080)) / np.float32(1920))
    assert abs(cam.getWorldStep() - 1 / 1440) < 1e-9
    np.testing.assert_array_equal(cam.getCameraPosition(), [1, 1, 6])
    np.testing.assert_array_equal(cam.getCameraDirection(), [0, 0, -1])
    np.testing.assert_array_equal(cam.getUpDirection(), [0, 1, 0])
    np.testing.assert_array_equal(cam.getRightDirection(), [1, 0, 0])
    ww, wh = np.float32(1920 / math.sqrt(1920 * 1080)), np.float32(1080 / math.sqrt(1920 * 1080))
    np.testing.assert_array_equal(cam.getUpperLeft(), np.array([-ww / 2 + 1, wh / 2 + 1, -1 + 6], dtype=np.float32))
    assert cam.data[3] == 0 and cam.data[7] == 0 and cam.data[11] == 0 and cam.data[15] == 0 and cam.data[20] == 0


@pytest.mark.parametrize("w,h,pos", [(1080, 720, (1, 1, 6)), (512, 512, (1, 0, 6)), (3840, 2160, (1, 0, 6)), (64, 36, (0, 0, 0))])
def test_camera_matches_oracle(rta, orc, w, h, pos):
    np.testing.assert_array_equal(rta.Camera(w, h, 1.0, pos).data.view(np.uint32),
                                  orc.camera_init(w, h, 1.0, pos).view(np.uint32))


def test_camera_move_and_rotate(rta):
    cam = rta.Camera(640, 480)
    ul0 = cam.getUpperLeft().copy()
    cam.move(rta.Camera.FORWARD)                     # camera.h:98: position += direction * 0.2
    np.testing.assert_allclose(cam.getCameraPosition(), [1, 1, 5.8], rtol=0, atol=1e-6)
    np.testing.assert_allclose(cam.getUpperLeft() - ul0, [0, 0, -0.2], atol=1e-6)
    cam.move(rta.Camera.LEFT)
    np.testing.assert_allclose(cam.getCameraPosition(), [0.8, 1, 5.8], atol=1e-6)
    cam.rotate(math.pi / 2, 0.0)                     # quarter turn about up: direction -z -> -x
    np.testing.assert_allclose(cam.getCameraDirection(), [-1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(cam.getRightDirection(), [0, 0, -1], atol=1e-6)
    np.testing.assert_allclose(cam.getUpDirection(), [0, 1, 0], atol=1e-6)
    for v in (cam.getCameraDirection(), cam.getRightDirection(), cam.getUpDirection()):
        assert abs(np.linalg.norm(v) - 1) < 1e-6
    # upper-left stays focal length ahead and half an image up/left of the eye (camera.h:91)
    ww, wh = 640 / math.sqrt(640 * 480), 480 / math.sqrt(640 * 480)
    expect = cam.getCameraPosition() + cam.getCameraDirection() - ww / 2 * cam.getRightDirection() + wh / 2 * cam.getUpDirection()
    np.testing.assert_allclose(cam.getUpperLeft(), expect, atol=1e-6)


def test_circle_and_box_layout(rta):
    c = rta.Circle((1.5, 1, -3), 1.1, rta.Lambertian((1, 0, 0)))
    t = c.transformation
    assert t[0, 0] == np.float32(1.1) and t[1, 1] == 1 and t[2, 2] == 1 and t[3, 3] == 1
    np.testing.assert_array_equal(t[:3, 3], np.array([1.5, 1, -3], dtype=np.float32))
    assert int(c.rec["type"]) == 0 and int(c.rec["material"]["type"]) == 1
    b = rta.Box((3, 2, -4.5), (1, 2, 1), rta.Light((1.5, 1.5, 1.5)))
    t = b.transformation
    np.testing.assert_array_equal(np.diag(t), [1, 2, 1, 1])
    np.testing.assert_array_equal(t[:3, 3], np.array([3, 2, -4.5], dtype=np.float32))
    assert int(b.rec["type"]) == 1 and int(b.rec["material"]["type"]) == 0
    m = rta.Metal((1, 1, 1))
    assert m.type == 2 and m.rec["data"][0] == 1.0          # shape.h:30-33: data[0] = refraction = 1
    assert rta.DebugMaterial((0, 0, 0)).type == 100


def test_simple_triangle_identities(rta, orc):
    rng = np.random.RandomState(3)
    for _ in range(50):
        p1, p2, p3 = rng.uniform(-2, 2, (3, 3)).astype(np.float32)
        t = rta.SimpleTriangle(p1, p2, p3).rec
        v1, v2 = (p2 - p1).astype(np.float64), (p3 - p1).astype(np.float64)
        a, m1, m2 = t["a"].astype(np.float64), t["m1"][:3].astype(np.float64), t["m2"][:3].astype(np.float64)
        assert abs(np.linalg.norm(a) - 1) < 1e-6
        assert abs(a @ v1) < 1e-5 and abs(a @ v2) < 1e-5
        assert abs(float(t["beta"]) - a @ p1) < 1e-5
        # dual basis: m1.v1 = 1, m1.v2 = 0, m2.v2 = 1, m2.v1 = 0 (shape.h:93-97)
        assert abs(m1 @ v1 - 1) < 1e-4 and abs(m1 @ v2) < 1e-4
        assert abs(m2 @ v2 - 1) < 1e-4 and abs(m2 @ v1) < 1e-4
        # centroid has barycentrics (1/3, 1/3)
        q = (p1 + p2 + p3).astype(np.float64) / 3 - p1
        assert abs(q @ m1 - 1 / 3) < 1e-4 and abs(q @ m2 - 1 / 3) < 1e-4
        o = orc.make_triangle(p1, p2, p3)
        assert t.tobytes() == o.tobytes()


def test_mesh_loader_counts(rta):
    for name, nv, nt, coff in (("Bunny-LowPoly.off", 148, 292, False), ("bunny.off", 2503, 4968, True)):
        m = rta.SimpleMesh()
        assert m.loadMesh(os.path.join(rta.scenes.MESH_DIR, name))
        pos, col = m.getVertices()
        idx = m.getTriangles()
        assert pos.shape == (nv, 4) and idx.shape == (nt, 3)
        assert np.all(pos[:, 3] == 1.0) and idx.max() < nv
        if coff:        # SimpleMesh.h:179: colours / 255; first vertex "0 255 0 0"
            np.testing.assert_array_equal(col[0], [0, 1, 0, 0])
        else:           # SimpleMesh.h:191-194
            np.testing.assert_array_equal(col[0], [0, 0, 0, 255])
    assert not rta.SimpleMesh().loadMesh("/nonexistent.off")
    first = np.array([29.22699, 18.70715, 44.53949], dtype=np.float32)      # Bunny-LowPoly.off:3
    m = rta.SimpleMesh(); m.loadMesh(os.path.join(rta.scenes.MESH_DIR, "Bunny-LowPoly.off"))
    np.testing.assert_array_equal(m.getVertices()[0][0, :3], first)


def test_mesh_normalize_resize_transform(rta):
    m = rta.SimpleMesh(); m.loadMesh(os.path.join(rta.scenes.MESH_DIR, "Bunny-LowPoly.off"))
    m.normalize()
    pos, _ = m.getVertices()
    assert np.abs(pos[:, :3].astype(np.float64).mean(0)).max() < 1e-5         # centred
    assert abs(np.linalg.norm(pos[:, :3], axis=1).max() - 1) < 1e-5          # max radius 1
    m.resize()
    pos2, _ = m.getVertices()
    np.testing.assert_array_equal(pos2, pos * 2)                              # incl. w (SimpleMesh.h:83)
    rot = np.eye(3, dtype=np.float32) * 0.5
    m.transform(rot, (0.5, 0.7, -2))
    pos3, _ = m.getVertices()
    np.testing.assert_allclose(pos3[:, :3], pos2[:, :3] * 0.5 + np.array([0.5, 0.7, -2], np.float32), atol=1e-6)


@pytest.mark.parametrize("name", ["Bunny-LowPoly.off", "bunny.off"])
def test_mesh_pipeline_matches_oracle(rta, orc, name):
    """main.cpp:108-122 through the library == the oracle's restatement, byte for byte."""
    tris = rta.scenes.main_cpp_mesh(name)
    otris, nv = orc.mesh_triangles(os.path.join(rta.scenes.MESH_DIR, name))
    assert len(tris) == len(otris)
    assert tris.tobytes() == otris.tobytes()
    # the bunny sits where main.cpp:112 puts it
    c = tris["position"][:, :3].mean(0)
    assert np.linalg.norm(c - np.array([0.5, 0.7, -2])) < 0.5


def test_scene_lists(rta):
    s = rta.scenes.main_cpp_shapes()                  # main.cpp:95-106
    assert len(s) == 9
    assert list(s["type"]) == [1, 0, 0, 0, 0, 1, 1, 0, 1]
    assert list(s["material"]["type"]) == [0, 1, 2, 1, 1, 1, 1, 1, 1]
    c = rta.scenes.cpu_sphere_scene()                 # rayTracer.h:48-54
    assert len(c) == 4 and list(c["material"]["type"]) == [1, 1, 1, 0]
    assert c["transformation"][2][0] == 100 and tuple(c["transformation"][3][12:15]) == (1, 2, -2)


def test_mesh_sphere_is_the_icosahedron(rta):
    # SimpleMesh::sphere, SimpleMesh.h:293-317: 12 vertices at distance scale from the centre, 20 faces,
    # every vertex used by five faces, every edge shared by two faces with opposite orientation.
    m = rta.SimpleMesh.sphere((1.0, 2.0, 3.0), 0.5, (0.25, 0.5, 0.75, 1.0))
    pos, col = m.getVertices()
    idx = m.getTriangles()
    assert pos.shape == (12, 4) and idx.shape == (20, 3)
    np.testing.assert_allclose(np.linalg.norm(pos[:, :3] - [1, 2, 3], axis=1), 0.5, rtol=1e-5)
    np.testing.assert_array_equal(pos[:, 3], 1.0)
    np.testing.assert_array_equal(col, np.tile(np.float32([0.25, 0.5, 0.75, 1.0]), (12, 1)))
    np.testing.assert_array_equal(np.bincount(idx.ravel(), minlength=12), 5)
    edges = {}
    for a, b, c in idx.tolist():
        for e in ((a, b), (b, c), (c, a)):
            edges[e] = edges.get(e, 0) + 1
    assert len(edges) == 60 and all(v == 1 and (e[1], e[0]) in edges for e, v in edges.items())
    np.testing.assert_array_equal(idx[0], [0, 4, 1])                     # SimpleMesh.h:300
    np.testing.assert_allclose(pos[0, :3], [1 - 0.5 * 0.525731, 2, 3 + 0.5 * 0.850651], rtol=1e-6)   # :298


def test_mesh_write_then_load(rta, tmp_path):
    # writeMesh, SimpleMesh.h:217-245: COFF header, "x y z r g b a" with int() colours, "3 i j k" faces.
    m = rta.SimpleMesh()
    m.addVertex((0, 0, 0, 1), (1, 0, 0, 1))
    m.addVertex((1.5, 0, 0, 1), (0, 2.9, 0, 1))
    m.addVertex((0, 0.25, 0, 1), (0, 0, 255, 255))
    m.addVertex((float("nan"), 0, 0, 1), (9, 9, 9, 9))                  # :229-233: non-finite -> zeros
    assert m.addFace(0, 1, 2) == 0 and m.addFace(2, 1, 0) == 1
    early = rta.SimpleMesh()                                           # a face before its vertices, as SimpleMesh::addFace allows (:99-104)
    assert early.addFace(0, 1, 2) == 0
    with pytest.raises(IndexError):
        early.toSimpleTriangles()                                      # checked where the indices are followed
    for p in ((0, 0, 0, 1), (1, 0, 0, 1), (0, 1, 0, 1)):
        early.addVertex(p)
    assert len(early.toSimpleTriangles()) == 1 and early.triangleColors().shape == (1, 3)
    path = tmp_path / "out.off"
    assert m.writeMesh(path)
    assert path.read_text().split("\n") == ["COFF", "4 2 0", "0 0 0 1 0 0 1", "1.5 0 0 0 2 0 1", "0 0.25 0 0 0 255 255",
                                             "0.0 0.0 0.0 0 0 0 0", "3 0 1 2", "3 2 1 0", ""]
    back = rta.SimpleMesh()
    assert back.loadMesh(path)
    pos, col = back.getVertices()
    np.testing.assert_array_equal(pos[:3, :3], [[0, 0, 0], [1.5, 0, 0], [0, 0.25, 0]])
    np.testing.assert_array_equal(col[2], np.float32([0, 0, 255, 255]) / np.float32(255))   # loadMesh :179
    np.testing.assert_array_equal(back.getTriangles(), [[0, 1, 2], [2, 1, 0]])
    assert not m.writeMesh(tmp_path / "no_such_dir" / "x.off")             # :220
    assert not back.loadMesh(tmp_path / "missing.off") and len(back.getTriangles()) == 0   # :147-154


def test_mesh_join(rta):
    # joinMeshes, SimpleMesh.h:251-288: mesh1 transformed by pose1to2, mesh2's indices offset.
    a = rta.SimpleMesh.sphere((0, 0, 0), 1.0, (1, 0, 0, 1))
    b = rta.SimpleMesh.sphere((5, 0, 0), 2.0, (0, 1, 0, 1))
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    pose[:3, 3] = [10, 20, 30]
    j = rta.SimpleMesh.joinMeshes(a, b, pose)
    pa, ca = a.getVertices()
    pb, cb = b.getVertices()
    pj, cj = j.getVertices()
    np.testing.assert_allclose(pj[:12], pa @ pose.T, atol=1e-6)
    np.testing.assert_array_equal(pj[12:], pb)
    np.testing.assert_array_equal(cj, np.concatenate([ca, cb]))
    np.testing.assert_array_equal(j.getTriangles(), np.concatenate([a.getTriangles(), b.getTriangles() + 12]))
    same = rta.SimpleMesh.joinMeshes(a, b)
    np.testing.assert_array_equal(same.getVertices()[0][:12], pa)


def test_mesh_colors_from_coff(rta, tmp_path):
    # loadMesh of a COFF file keeps colour / 255 per vertex (SimpleMesh.h:170-183); a face's colour is
    # the mean of its three vertices'.  setColor overwrites all (:87-91).
    path = tmp_path / "c.off"
    path.write_text("COFF\n4 2 0\n0 0 0 255 0 0 255\n1 0 0 0 255 0 255\n0 1 0 0 0 255 255\n0 0 1 51 102 153 255\n"
                    "3 0 1 2\n3 1 2 3\n")
    m = rta.SimpleMesh()
    assert m.loadMesh(path)
    col = m.triangleColors()
    third = np.float32(1) / np.float32(3)
    np.testing.assert_allclose(col[0], [third, third, third], rtol=1e-6)
    np.testing.assert_allclose(col[1], [51 / 255 / 3, (255 + 102) / 255 / 3, (255 + 153) / 255 / 3], rtol=1e-6)
    m.setColor(0.5, 0.25, 0.125, 1.0)
    np.testing.assert_array_equal(m.triangleColors(), np.tile(np.float32([0.5, 0.25, 0.125]), (2, 1)))
    m.clear()
    assert len(m.getTriangles()) == 0 and m.getVertices()[0].shape == (0, 4)
