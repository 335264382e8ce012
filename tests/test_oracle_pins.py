"""Pins the CPU oracle (oracle/rt_oracle.c) to the reference.

Three kinds of pin, strongest first:
  1. outputs the UNMODIFIED reference Renderer produced in the build container (accumulation-buffer checksums and the
     TraceRay call count recorded by the survey run) -- the whole PerPixel/TraceRay/shading/accumulate chain;
  2. golden fixtures generated from the reference's own objects (tests/golden/make_golden.py: KDTreeCPU, Intersections,
     Camera, Walnut::Random/pcg32, glm compiled unmodified into oracle/_ref);
  3. live comparison against oracle/_ref where it exists (the build container; skipped elsewhere).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import bits, default_sphere_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def gold(name):
    return np.load(os.path.join(GOLD, name))


# ---- 1. the reference Renderer's recorded outputs ---------------------------------------------------------------------
@pytest.mark.parametrize("w,h,frames,fnv,rays", [
    (320, 180, 4, 0xcb022dd9, None),
    (1280, 720, 10, 0x7ee58aae, 33805972),
])
def test_reference_renderer_checksums(orc, w, h, frames, fnv, rays):
    """Default sphere scene, 5 bounces, one default-seeded pcg32 stream in row-major order (what the reference does when
    its row loop runs serially), glibc trig through the double overloads (how g++ resolves Renderer.h:50-52's unqualified
    sqrt/cos/sin): the accumulation buffer after `frames` Render calls hashes (FNV-1a over 32-bit words) to the value the
    unmodified reference produced, and the 1280x720 run makes exactly the 33,805,972 TraceRay calls gprof counted."""
    sc = default_sphere_scene(orc)
    cam = orc.camera(70.0, 0.05, 100.0, w, h)
    st = orc.Settings(1, 1, 1, 0, 5)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    cnt = orc.Counters()
    orc.serial_rng_reset()
    for f in range(1, frames + 1):
        sc.render_frame(st, cam["pos"], cam["dirs"], w, h, f, acc, img, rng_mode=orc.RNG_SERIAL_LIBM_DOUBLE, counters=cnt)
    assert orc.fnv1a(acc) == fnv
    if rays is not None:
        assert cnt.rays == rays


def test_survey_known_answers(orc):
    """Values printed by the unmodified reference headers/objects in the survey run."""
    L = orc.lib()
    st = np.array([L.orc_pcg32_seed(0xcafef00dd15ea5e5)], np.uint64)
    got = [L.orc_pcg32_next(p(st)) for _ in range(6)]
    assert got == [0x285594ea, 0x190ca349, 0xcbc42ff2, 0xd6508153, 0xc2a8052f, 0x0f55ac5f]
    st = np.array([L.orc_pcg32_seed(0xcafef00dd15ea5e5)], np.uint64)
    fl = np.array([L.orc_u32_to_float(L.orc_pcg32_next(p(st))) for _ in range(4)], np.float32)
    assert list(bits(fl)) == [0x3e215654, 0x3dc8651a, 0x3f4bc430, 0x3f565081]
    cam = orc.camera(70.0, 0.05, 100.0, 1280, 720)
    assert list(bits(cam["fwd"])) == [0x00000000, 0xbdcbc89b, 0xbf7ebac1]
    d = cam["dirs"].reshape(720, 1280, 3)
    assert list(bits(d[0, 0])) == [0xbf36c68a, 0xbee9d27d, 0xbf07def3]
    assert list(bits(d[0, 1279])) == [0x3f36a2a8, 0xbeea0230, 0xbf07faaa]
    assert list(bits(d[360, 640])) == [0x00000000, 0xbdcbc89a, 0xbf7ebac2]
    assert list(bits(d[719, 1279])) == [0x3f36b79e, 0x3eaf07ff, 0xbf1c7f52]
    assert np.all(d[:, 640, 0] == 0.0)          # the centre column has direction.x == 0 exactly: the NaN slab case is live


# ---- 2. golden fixtures from the reference's own objects -----------------------------------------------------------------
def test_rng_golden(orc):
    g = gold("ref_kat.npz")
    L = orc.lib()
    st = np.array([L.orc_pcg32_seed(0xcafef00dd15ea5e5)], np.uint64)
    assert [L.orc_pcg32_next(p(st)) for _ in range(32)] == list(g["pcg32_default"])
    st = np.array([L.orc_pcg32_seed(12345)], np.uint64)
    assert [L.orc_pcg32_next(p(st)) for _ in range(32)] == list(g["pcg32_seed_12345"])
    st = np.array([L.orc_pcg32_seed(0xcafef00dd15ea5e5)], np.uint64)
    fl = np.array([L.orc_u32_to_float(L.orc_pcg32_next(p(st))) for _ in range(32)], np.float32)
    assert np.array_equal(bits(fl), g["random_float_bits"])


@pytest.mark.parametrize("w,h", [(64, 36), (320, 180), (1280, 720), (1920, 1080), (3840, 2160)])
def test_camera_golden(orc, w, h):
    g = gold("ref_kat.npz")
    tag = "cam_%dx%d_" % (w, h)
    cam = orc.camera(70.0, 0.05, 100.0, w, h)
    for k in ("pos", "fwd", "proj", "inv_proj", "view", "inv_view"):
        assert np.array_equal(bits(cam[k]), bits(g[tag + k])), k
    d = cam["dirs"].reshape(h, w, 3)
    assert orc.fnv1a(cam["dirs"]) == int(g[tag + "dirs_fnv"][0])
    corners = np.stack([d[0, 0], d[0, w - 1], d[h // 2, w // 2], d[h - 1, w - 1]])
    assert np.array_equal(bits(corners), bits(g[tag + "corners"]))
    assert np.array_equal(bits(d[:, w // 2, 0]), bits(g[tag + "centre_column_x"]))
    if w == 64:
        assert np.array_equal(bits(cam["dirs"]), bits(g[tag + "dirs"]))


def test_glm_primitives_golden(orc):
    g = gold("ref_glm.npz")
    L = orc.lib()
    a, b, s = g["a"], g["b"], g["s"]
    o = np.zeros(3, np.float32)
    for i in range(len(a)):
        ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
        L.orc_glm_normalize(p(ai), p(o)); assert np.array_equal(bits(o), bits(g["normalize"][i]))
        L.orc_glm_reflect(p(ai), p(bi), p(o)); assert np.array_equal(bits(o), bits(g["reflect"][i]))
        L.orc_glm_mix(p(ai), p(bi), float(s[i]), p(o)); assert np.array_equal(bits(o), bits(g["mix"][i]))
        L.orc_glm_cross(p(ai), p(bi), p(o)); assert np.array_equal(bits(o), bits(g["cross"][i]))
        d = np.array([L.orc_glm_dot(p(ai), p(bi))], np.float32)
        assert bits(d)[0] == bits(g["dot"][i:i + 1])[0]
    inv = np.zeros(16, np.float32); mv = np.zeros(4, np.float32)
    for i in range(len(g["mat"])):
        m, v = np.ascontiguousarray(g["mat"][i]), np.ascontiguousarray(g["vec4"][i])
        L.orc_glm_inverse(p(m), p(inv)); assert np.array_equal(bits(inv), bits(g["inverse"][i]))
        L.orc_glm_mat_vec(p(m), p(v), p(mv)); assert np.array_equal(bits(mv), bits(g["mat_vec"][i]))


def test_tonemap_golden(orc):
    """ACESFilm + LinearToSRGB through real glm (glibc powf) vs the oracle's libm path (same libm: a few ulp at most)
    and vs the portable pow the GPU shares (display-only difference, bounded)."""
    g = gold("ref_glm.npz")
    L = orc.lib()
    cin, want = g["tonemap_in"], g["tonemap_glibc_powf"]
    out = np.zeros(3, np.float32)
    worst_port = 0.0
    for i in range(len(cin)):
        c = np.ascontiguousarray(cin[i])
        L.orc_tonemap(p(c), 1, 0, p(out))
        assert np.max(np.abs(out.astype(np.float64) - want[i])) <= 2.5e-7, (i, out, want[i])
        L.orc_tonemap(p(c), 1, 1, p(out))
        worst_port = max(worst_port, float(np.max(np.abs(out.astype(np.float64) - want[i]))))
    assert worst_port <= 1.0e-6
    z = np.zeros(3, np.float32)
    L.orc_tonemap(p(z), 1, 1, p(out))
    assert np.array_equal(bits(out), bits(z))                        # black stays +0


def test_intersections_golden(orc):
    g = gold("ref_intersections.npz")
    L = orc.lib()
    n = len(g["center"])
    tn = np.zeros(1, np.float32)
    flt_max = np.float32(3.4028234663852886e38)
    nan_hits = 0
    for i in range(n):
        tn[0] = flt_max
        args = [np.ascontiguousarray(g[k][i]) for k in ("center", "extent", "origin", "inv_d")]
        hit = L.orc_aabb(*[p(a) for a in args], p(tn))
        assert hit == g["aabb_hit"][i], i
        assert bits(tn)[0] == bits(g["aabb_tnear"][i:i + 1])[0], i
        nan_hits += int(hit and np.isnan(tn[0]))
    assert nan_hits > 0                  # the fixture does exercise "NaN slab counts as a hit with t_near = NaN"
    t = np.zeros(1, np.float32); u = np.zeros(1, np.float32); v = np.zeros(1, np.float32)
    hits = 0
    for i in range(n):
        t[0] = u[0] = v[0] = 0.0
        args = [np.ascontiguousarray(g[k][i]) for k in ("tri_origin", "tri_dir", "v0", "v1", "v2")]
        hit = L.orc_tri(*[p(a) for a in args], p(t), p(u), p(v))
        assert hit == g["tri_hit"][i], i
        if hit:
            hits += 1
            assert (bits(t)[0], bits(u)[0], bits(v)[0]) == (bits(g["tri_t"][i:i + 1])[0], bits(g["tri_u"][i:i + 1])[0], bits(g["tri_v"][i:i + 1])[0])
    assert 200 < hits < n - 200


def _mesh(name):
    from raytracer_amd import scenes
    return scenes.displaced_grid(40) if name == "grid40" else scenes.cube_sphere(16)


def _leaf_lists(flat):
    leaves = (flat["left"] < 0) & (flat["right"] < 0)
    return leaves, [tuple(flat["leaf_idx"][f:f + c]) for f, c in zip(flat["first"][leaves], flat["count"][leaves])]


@pytest.mark.parametrize("name", ["grid40", "cubesphere16"])
def test_kd_build_and_traversal_golden(orc, name):
    """The oracle's restated builder grows the tree the reference's KDTreeCPU grew (same boxes, children, leaf triangle
    order) and its traversal returns the reference's hits bit for bit, for camera rays and for rays from surface points."""
    g = gold("ref_kd_%s.npz" % name)
    verts, tris, normals9, tri_mat, _ = _mesh(name)
    assert (orc.fnv1a(verts), orc.fnv1a(tris)) == tuple(int(x) for x in g["mesh_fnv"]), "mesh generator drifted from the fixture"
    sc = orc.Scene()
    sc.set_mesh(verts, tris, normals9, tri_mat)
    sc.build_kd()
    mine = sc.export_kd()
    assert mine["depth"] == int(g["depth"][0])
    for k in ("left", "right"):
        assert np.array_equal(mine[k], g[k]), k
    for k in ("center", "extent"):
        assert np.array_equal(bits(mine[k]), bits(g[k])), k
    gl = dict(left=g["left"], right=g["right"], first=g["first"], count=g["count"], leaf_idx=g["leaf_idx"])
    assert _leaf_lists(mine)[1] == _leaf_lists(gl)[1]

    w, h = (int(x) for x in g["width"])
    dirs = orc.camera(70.0, 0.05, 100.0, w, h)["dirs"]
    ids, t, u, v, _ = sc.trace_primary(0, g["cam_pos"], dirs)
    assert np.array_equal(ids, g["primary_ids"])
    hit = ids >= 0
    assert hit.sum() > 500
    for a, k in ((t, "primary_t"), (u, "primary_u"), (v, "primary_v")):
        assert np.array_equal(bits(a[hit]), bits(g[k][hit])), k
    # rays leaving surface points: one call per ray (own origin)
    for i in range(0, len(g["ray_o"]), 7):
        ids, t, u, v, _ = sc.trace_primary(0, g["ray_o"][i], g["ray_d"][i:i + 1])
        assert ids[0] == g["ray_ids"][i]
        if ids[0] >= 0:
            assert (bits(t)[0], bits(u)[0], bits(v)[0]) == (bits(g["ray_t"][i:i + 1])[0], bits(g["ray_u"][i:i + 1])[0], bits(g["ray_v"][i:i + 1])[0])


# ---- 3. live against the reference's own objects ---------------------------------------------------------------------------
def test_live_kd_against_reference_objects(orc, ref):
    from raytracer_amd import scenes
    verts, tris, normals9, tri_mat, _ = scenes.displaced_grid(23, drop=5)
    kd = ref.KDTree(verts, tris)
    theirs = kd.export()
    sc = orc.Scene(); sc.set_mesh(verts, tris, normals9, tri_mat); sc.build_kd()
    mine = sc.export_kd()
    for k in ("left", "right", "first", "count", "leaf_idx"):
        assert np.array_equal(mine[k], theirs[k]), k
    for k in ("center", "extent"):
        assert np.array_equal(bits(mine[k]), bits(theirs[k])), k
    cam = orc.camera(70.0, 0.05, 100.0, 96, 54)
    a = kd.intersect(cam["pos"], cam["dirs"])
    b = sc.trace_primary(0, cam["pos"], cam["dirs"])
    assert np.array_equal(a[0], b[0])
    hit = a[0] >= 0
    for i in (1, 2, 3):
        assert np.array_equal(bits(a[i][hit]), bits(b[i][hit]))


@pytest.mark.parametrize("log2_scale", [33, 45, 61])
def test_live_kd_at_absurd_scale_against_reference_objects(orc, ref, log2_scale):
    """KDTreeCPU::intersect reports a hit as soon as any triangle test succeeds, also one whose distance overflowed to +inf and
    was never accepted: index, t, u, v then keep the caller's initial values (KDTreeCPU.cpp:666-690).  From about 2^40 units on
    that happens; the oracle must restate it (the library's walks are compared with the oracle there on the GPU)."""
    from raytracer_amd import scenes
    k = np.float32(2.0 ** log2_scale)
    centre = np.array([0.0, 0.6, -0.8], np.float32)
    verts, tris, normals9, tri_mat, _ = scenes.cube_sphere(10, center=tuple(centre), radius=0.7)
    verts = (verts * k).astype(np.float32)
    kd = ref.KDTree(verts, tris)
    sc = orc.Scene(); sc.set_mesh(verts, tris, normals9, tri_mat); sc.build_kd()
    theirs, mine = kd.export(), sc.export_kd()
    for key in ("left", "right", "first", "count", "leaf_idx"):
        assert np.array_equal(mine[key], theirs[key]), key
    d = np.array((0.3, 0.5, 1.0), np.float32); d /= np.linalg.norm(d)
    cam = orc.camera(70.0, 0.05, 100.0, 96, 64, pos=tuple(centre + np.float32(1.9) * d), fwd=tuple(-d))
    pos = ((centre + np.float32(1.9) * d) * k).astype(np.float32)
    a = kd.intersect(pos, cam["dirs"])
    b = sc.trace_primary(0, pos, cam["dirs"])
    assert np.array_equal(a[0], b[0])
    hit = a[0] >= 0
    assert hit.sum() > 500
    for i in (1, 2, 3):
        assert np.array_equal(bits(a[i][hit]), bits(b[i][hit]))
    if log2_scale == 61:                                     # every test that succeeds does so at t = +inf: triangle 0, nothing accepted
        assert (a[0][hit] == 0x40000000).all()
    if log2_scale == 33:
        assert len(np.unique(a[0][hit])) > 100


def test_live_camera_and_rng_against_reference_objects(orc, ref):
    for (w, h) in [(33, 17), (640, 360)]:
        a = ref.camera(70.0, 0.05, 100.0, w, h); b = orc.camera(70.0, 0.05, 100.0, w, h)
        for k in ("pos", "fwd", "proj", "inv_proj", "view", "inv_view", "dirs"):
            assert np.array_equal(bits(a[k]), bits(b[k])), (w, h, k)
    L = orc.lib()
    st = np.array([L.orc_pcg32_seed(987654321)], np.uint64)
    assert [L.orc_pcg32_next(p(st)) for _ in range(64)] == list(ref.pcg32_u32(64, seed=987654321))


# ---- the two portable functions the GPU shares ----------------------------------------------------------------------------
def test_portable_sincos_accuracy(orc):
    L = orc.lib()
    s = np.zeros(1, np.float32); c = np.zeros(1, np.float32)
    worst = 0.0
    for a in np.linspace(0.0, 2.0 * np.pi, 20001, dtype=np.float32):
        L.orc_sincos(float(a), p(s), p(c))
        worst = max(worst, abs(float(s[0]) - np.sin(np.float64(a))), abs(float(c[0]) - np.cos(np.float64(a))))
    assert worst < 2.0e-7


def test_portable_pow_accuracy(orc):
    L = orc.lib()
    worst = 0.0
    for x in np.linspace(0.0031308, 1.0, 20001, dtype=np.float32):
        got = L.orc_pow_inv24(float(x))
        want = np.float64(x) ** (1.0 / 2.4)
        worst = max(worst, abs(got - want) / want)
    assert worst < 6.0e-7
    assert np.isfinite(L.orc_pow_inv24(0.0)) and np.isfinite(L.orc_pow_inv24(1e-30))
