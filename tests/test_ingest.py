"""OBJ/MTL ingest with the reference application's conventions (RaytracerApp.cpp:27-131): data checks on the CPU, one GPU
render of an ingested scene against the oracle."""
import numpy as np
import pytest

from conftest import bits

OBJ = """# a unit cube on a quad floor; the cube's top is 'water'
mtllib box.mtl
v -2 0 -3
v  2 0 -3
v  2 0  1
v -2 0  1
v -0.5 0.0 -1.5
v  0.5 0.0 -1.5
v  0.5 0.0 -0.5
v -0.5 0.0 -0.5
v -0.5 1.0 -1.5
v  0.5 1.0 -1.5
v  0.5 1.0 -0.5
v -0.5 1.0 -0.5
vn 0 2 0
vn 0 0 3
vn 1 0 0
f 1//1 4//1 3//1 2//1
usemtl side
f 8//2 7//2 11//2 12//2
f 7//3 6//3 10//3 11//3
f 5 8 12 9
f -7 -8 -4 -3
usemtl water
f 9//1 12//1 11//1 10//1
usemtl lamp
f 1 2 10
"""
MTL = """newmtl side
Kd 0.2 0.7 0.1
Ks 0.5 0.1 0.1
Ns 512
newmtl water
Kd 0.1 0.2 0.9
Ni 1.33
newmtl lamp
Ke 2 2 1.5
"""


@pytest.fixture
def box(tmp_path):
    (tmp_path / "box.obj").write_text(OBJ)
    (tmp_path / "box.mtl").write_text(MTL)
    from raytracer_amd import ingest
    return ingest.load_obj(str(tmp_path / "box.obj"), app_materials=True)


def test_obj_ingest_follows_the_reference_application(box):
    verts, tri_idx, normals9, tri_mat, materials, names = box
    T = 13                                             # six quads -> twelve triangles, plus one triangle
    assert len(tri_idx) == T and len(verts) == 3 * T and normals9.shape == (T, 9)
    assert np.array_equal(tri_idx.ravel(), np.arange(3 * T))                        # every triangle owns its vertices
    assert names == ["side", "water", "lamp"] and len(materials) == 3 + 7           # the application's seven come after
    # faces before any usemtl take material 0; then side, water, lamp
    assert tri_mat.tolist() == [0, 0] + [0] * 8 + [1, 1] + [2]
    side, water, lamp = materials[0], materials[1], materials[2]
    assert np.array_equal(side["albedo"], np.array([0.5, 0.7, 0.1], np.float32))    # max(Kd, Ks) per channel
    assert side["roughness"] == np.float32(0.5) and side["transparency"] == 0.0
    assert water["transparency"] == 1.0 and water["ior"] == np.float32(1.33) and water["roughness"] == np.float32(1023.0 / 1024.0)
    assert np.array_equal(lamp["emission"], np.array([4.0, 4.0, 3.0], np.float32))  # 2 * Ke
    # normals are normalised per corner; corners without one get (0, 1, 0)
    assert np.array_equal(normals9[0], np.tile([0.0, 1.0, 0.0], 3).astype(np.float32))
    assert np.array_equal(normals9[2, :3], np.array([0.0, 0.0, 1.0], np.float32))
    assert np.array_equal(normals9[6], np.tile([0.0, 1.0, 0.0], 3).astype(np.float32))
    # the floor quad is a square: equal diagonals -> the "else" split [0,1,3], [1,2,3] of the loader
    floor = verts[:6].reshape(2, 3, 3)
    v = np.array([[-2, 0, -3], [-2, 0, 1], [2, 0, 1], [2, 0, -3]], np.float32)      # face order 1 4 3 2
    assert np.array_equal(floor[0], v[[0, 1, 3]]) and np.array_equal(floor[1], v[[1, 2, 3]])


@pytest.mark.gpu
def test_ingested_scene_renders_like_the_oracle(box, orc):
    import raytracer_amd as rt
    from raytracer_amd import scenes
    verts, tri_idx, normals9, tri_mat, materials, _ = box
    scene = rt.Scene()
    scene.spheres = scenes.default_spheres(mat_offset=3)                  # the application offsets its sphere materials
    scene.materials = materials
    scene.set_triangles(verts, tri_idx, normals9, tri_mat, max_nodes=1 << 16)
    osc = orc.Scene()
    osc.set_spheres(scene.spheres); osc.set_materials(scene.materials)
    osc.set_mesh(verts, tri_idx, normals9, tri_mat); osc.import_kd(scene.kd_tree.arrays())
    w, h, frames = 160, 120, 3
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    r = rt.Renderer(0)
    for mode in (0, 2):
        st = r.GetSettings(); st.SceneMode, st.Bounces = mode, 6
        r.OnResize(w, h); r.ResetFrameIndex(); r.record_ids(True)
        acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
        for f in range(1, frames + 1):
            r.Render(scene, cam)
            osc.render_frame(orc.Settings(1, mode, 1, 0, 6), cam.GetPosition(), dirs, w, h, f, acc, img, primary_ids=ids if f == 1 else None)
        assert np.array_equal(r.read_primary_ids().ravel(), ids)
        assert np.array_equal(bits(r.read_accum().reshape(-1, 3)), bits(acc))
        assert np.array_equal(bits(r.GetImage().data.reshape(-1, 4)), bits(img))
        assert (ids >= 0x40000000).sum() > 2000


def test_polygons_are_triangulated_like_the_reference_loader():
    """Polygons with five to eight corners, convex and concave, in four different planes: the triangles, their order and their
    corner order equal what the reference's own tiny_obj_loader.h returns (tests/golden/ref_obj_polygons.npz, generated from the
    unmodified header by tests/golden/make_golden_obj.py)."""
    import os
    from conftest import ROOT
    from raytracer_amd import ingest
    gold = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(gold, "ref_obj_polygons.npz"))
    verts, tri_idx, normals9, tri_mat, mats, names = ingest.load_obj(os.path.join(gold, "polygons.obj"))
    pos, nrm = [], []
    with open(os.path.join(gold, "polygons.obj")) as fh:
        for line in fh:
            tok = line.split("#", 1)[0].split()
            if tok and tok[0] == "v":
                pos.append([float(x) for x in tok[1:4]])
            elif tok and tok[0] == "vn":
                nrm.append([float(x) for x in tok[1:4]])
    pos, nrm = np.array(pos, np.float32), np.array(nrm, np.float32)
    assert len(tri_idx) == len(g["v_idx"]) == 26
    assert np.array_equal(verts, pos[g["v_idx"].reshape(-1)])              # same triangles, same order, same corner order
    want_n = np.where(g["n_idx"].reshape(-1, 1) >= 0, nrm[np.maximum(g["n_idx"].reshape(-1), 0)], np.array([0.0, 1.0, 0.0], np.float32))
    assert np.array_equal(normals9.reshape(-1, 3), want_n)                 # unit normals in the file: normalising leaves them
    assert not np.array_equal(g["v_idx"][17:19], [[19, 24, 25], [19, 25, 26]])   # (a fan from corner 0 would differ on the star)
    assert np.all(tri_mat == 0)

