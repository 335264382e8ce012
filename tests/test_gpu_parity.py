"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI against the CPU oracle on the same
scenes, camera, settings and per-(pixel, frame) random streams.

Bar: primary-hit ids bit-exact, accumulated radiance bit-exact (so also within the stated tolerance
|d| <= 1e-5 + 1e-5*|ref|), the tone-mapped RGBA32F frame bit-exact, ray counters equal.  Mesh scenes run in both tree
walk orders: in the reference order the node/triangle counters must equal the oracle's as well (same visited set ray by
ray); front to back must give the same ids, radiance and image with no more nodes and triangles.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, bits

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    return raytracer_amd


@pytest.fixture(scope="module")
def renderer(rt):
    return rt.Renderer(0)


def make_scene(rt, orc, mesh=None, spheres=True, mats_extra=None):
    """Product scene + oracle scene holding the same data (the oracle takes the product's flat tree: the two builders
    are proven equal by the CPU tests)."""
    from raytracer_amd import scenes
    scene = rt.Scene()
    mats = scenes.default_materials()
    if spheres:
        scene.spheres = scenes.default_spheres()
    osc = orc.Scene()
    if mesh is not None:
        verts, tris, normals9, tri_mat, mesh_mats = mesh
        tri_mat = (np.asarray(tri_mat, np.uint32) + np.uint32(len(mats))).astype(np.uint32)
        mats = np.concatenate([mats, mesh_mats])
        scene.set_triangles(verts, tris, normals9, tri_mat, nthreads=4)
        osc.set_mesh(verts, tris, normals9, tri_mat)
        osc.import_kd(scene.kd_tree.arrays())
    if mats_extra is not None:
        mats = np.concatenate([mats, mats_extra])
    scene.materials = mats
    osc.set_spheres(scene.spheres); osc.set_materials(scene.materials)
    return scene, osc


REF_ORDER, F2B_ORDER = 1, 2          # _capi.ORDER_REFERENCE, _capi.ORDER_FRONT_TO_BACK
ORDERS = [pytest.param(REF_ORDER, id="reference-order"), pytest.param(F2B_ORDER, id="front-to-back")]


def run_both(rt, orc, r, scene, osc, w, h, mode, bounces, frames, use_ace=True, aa=False, seed=1337, stats=False,
             cam=None, threads=8, order=0, reuse=False):
    if cam is None:
        cam = rt.Camera(70.0, 0.05, 100.0)
    cam.OnResize(w, h)
    st = r.GetSettings()
    st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed = mode, bounces, use_ace, aa, True, seed
    st.TraversalOrder, st.ReuseFirstHits = order, reuse
    r.OnResize(w, h)
    r.ResetFrameIndex()
    r.record_ids(True)
    r.set_stats(stats)
    r.reset_counters()
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
    ost = orc.Settings(1, mode, int(use_ace), int(aa), bounces)
    cnt = orc.Counters()
    for f in range(1, frames + 1):
        r.Render(scene, cam)
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img, seed=seed, primary_ids=ids if f == 1 else None,
                         counters=cnt, nthreads=threads)
    got = dict(ids=r.read_primary_ids().ravel(), acc=r.read_accum().reshape(-1, 3), img=r.GetImage().data.reshape(-1, 4),
               cnt=r.read_counters())
    want = dict(ids=ids, acc=acc, img=img, cnt=cnt)
    r.set_stats(False)
    st.TraversalOrder, st.ReuseFirstHits = 0, False
    return got, want


def assert_exact(got, want, counters=("rays",)):
    assert np.array_equal(got["ids"], want["ids"]), "primary ids: %d pixels differ" % int((got["ids"] != want["ids"]).sum())
    assert np.array_equal(bits(got["acc"]), bits(want["acc"])), "accumulation: %d values differ, max |d| %g" % (
        int((bits(got["acc"]) != bits(want["acc"])).sum()), float(np.nanmax(np.abs(got["acc"] - want["acc"]))))
    # the stated tolerance (trivially met when bit-exact; kept so the bar is written down)
    assert np.all(np.abs(got["acc"] - want["acc"]) <= 1e-5 + 1e-5 * np.abs(want["acc"]))
    assert np.array_equal(bits(got["img"]), bits(want["img"])), "image differs"
    for k in counters:
        assert getattr(got["cnt"], k) == getattr(want["cnt"], k), (k, getattr(got["cnt"], k), getattr(want["cnt"], k))


def assert_mesh(got, want, order, extra=()):
    """Reference order: the same nodes and triangles as the oracle.  Front to back: same result, no more work."""
    if order == REF_ORDER:
        assert_exact(got, want, ("rays", "nodes_visited", "tris_tested") + tuple(extra))
    else:
        assert_exact(got, want, ("rays",) + tuple(extra))
        assert 0 < got["cnt"].nodes_visited <= want["cnt"].nodes_visited
        assert got["cnt"].tris_tested <= want["cnt"].tris_tested


# ---- spheres --------------------------------------------------------------------------------------------------------------
def test_default_sphere_scene_bit_exact(rt, orc, renderer):
    scene, osc = make_scene(rt, orc)
    got, want = run_both(rt, orc, renderer, scene, osc, 320, 180, 1, 5, 4)
    assert_exact(got, want, ("rays", "sphere_tests"))
    assert (want["ids"] >= 0).all() and len(np.unique(want["ids"])) >= 8       # the closed box: every pixel hits, most spheres seen


@pytest.mark.parametrize("w,h", [(1, 1), (17, 13), (16, 16), (333, 7), (8, 250)])
def test_ragged_frame_sizes(rt, orc, renderer, w, h):
    scene, osc = make_scene(rt, orc)
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, 1, 5, 2)
    assert_exact(got, want)


@pytest.mark.parametrize("bounces", [0, 1, 2, 9])
def test_bounce_counts(rt, orc, renderer, bounces):
    scene, osc = make_scene(rt, orc)
    got, want = run_both(rt, orc, renderer, scene, osc, 96, 64, 1, bounces, 2)
    assert_exact(got, want)
    if bounces == 0:
        assert got["cnt"].rays == 0 and not got["acc"].any()


def test_no_tonemap_and_other_seed(rt, orc, renderer):
    scene, osc = make_scene(rt, orc)
    got, want = run_both(rt, orc, renderer, scene, osc, 128, 72, 1, 5, 3, use_ace=False, seed=99)
    assert_exact(got, want)
    got2, _ = run_both(rt, orc, renderer, scene, osc, 128, 72, 1, 5, 3, use_ace=False, seed=1337)
    assert not np.array_equal(bits(got["acc"]), bits(got2["acc"]))             # the seed reaches the streams


def test_anti_aliasing_filter(rt, orc, renderer):
    scene, osc = make_scene(rt, orc)
    got, want = run_both(rt, orc, renderer, scene, osc, 150, 70, 1, 5, 3, aa=True)
    assert_exact(got, want)


def test_refraction_and_total_internal_reflection(rt, orc, renderer):
    """Glass spheres with a high index: exercises always-transmit, the inside/outside branches and the TIR fall-through."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc)
    scene.materials = scene.materials.copy()
    scene.materials[3] = scenes.make_material((0.9, 0.9, 1.0), roughness=0.2, transparency=1.0, ior=1.9)
    scene.materials[6] = scenes.make_material((1.0, 0.9, 0.9), roughness=0.0, transparency=0.5, ior=2.4)
    osc.set_materials(scene.materials)
    got, want = run_both(rt, orc, renderer, scene, osc, 200, 120, 1, 8, 3)
    assert_exact(got, want)


@pytest.mark.parametrize("extra,mode", [(40, 1), (150, 1), (150, 2)], ids=["49-spheres", "159-spheres", "159-spheres+mesh"])
def test_many_spheres_beyond_the_kernel_argument_and_lds_copies(rt, orc, renderer, extra, mode):
    """The first 32 spheres travel in the kernel arguments and scenes of up to 128 spheres / materials keep them in LDS;
    larger scenes read both from HBM (other kernel variants).  Default scene + a cloud of small spheres, each with a
    material of its own (diffuse, mirror-like, glass and emissive ones), bit-exact against the oracle."""
    from raytracer_amd import scenes
    rng = np.random.default_rng(7)
    mesh = None
    if mode == 2:
        mesh = scenes.displaced_grid(12, x_range=(-1.2, 1.2), z_range=(-2.4, 0.0), mat_offset=0)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    base = len(scene.materials)
    sph = np.zeros(extra, dtype=scenes.SPHERE_DTYPE)
    sph["x"] = rng.uniform(-0.95, 0.95, extra); sph["y"] = rng.uniform(0.05, 1.8, extra); sph["z"] = rng.uniform(-1.9, 0.5, extra)
    sph["radius"] = rng.uniform(0.03, 0.12, extra); sph["material"] = base + np.arange(extra)
    mats = []
    for i in range(extra):
        kind = i % 4
        col = tuple(rng.uniform(0.2, 1.0, 3))
        if kind == 0: mats.append(scenes.make_material(col, roughness=1.0))
        elif kind == 1: mats.append(scenes.make_material(col, roughness=0.05))
        elif kind == 2: mats.append(scenes.make_material(col, roughness=0.0, transparency=1.0, ior=1.3))
        else: mats.append(scenes.make_material(col, emission=tuple(rng.uniform(0.5, 3.0, 3)), roughness=1.0))
    scene.spheres = np.concatenate([scene.spheres, sph])
    scene.materials = np.concatenate([scene.materials, np.array(mats, dtype=scenes.MATERIAL_DTYPE)])
    osc.set_spheres(scene.spheres); osc.set_materials(scene.materials)
    got, want = run_both(rt, orc, renderer, scene, osc, 160, 100, mode, 6, 3, order=F2B_ORDER if mode == 2 else 0)
    assert_exact(got, want, ("rays", "sphere_tests"))
    assert len(np.unique(want["ids"][(want["ids"] >= 9) & (want["ids"] < 0x40000000)])) >= extra // 3     # the cloud is seen


def test_device_ray_generation_bit_exact(rt, orc, renderer):
    g = np.load(os.path.join(GOLD, "ref_kat.npz"))
    scene, _ = make_scene(rt, orc)
    for (w, h) in [(64, 36), (320, 180), (1920, 1080)]:
        cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
        renderer.OnResize(w, h)
        renderer.upload(scene, cam)
        d = renderer.read_ray_dirs()
        assert orc.fnv1a(d) == int(g["cam_%dx%d_dirs_fnv" % (w, h)][0])        # the reference Camera's own directions
        if w == 64:
            assert np.array_equal(bits(d), bits(g["cam_64x36_dirs"]))
    assert np.all(d.reshape(1080, 1920, 3)[:, 960, 0] == 0.0)


def test_host_supplied_ray_directions(rt, orc, renderer):
    scene, osc = make_scene(rt, orc)
    w, h = 120, 80
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, 1, 5, 2)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    renderer.set_ray_dirs(orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h))
    try:
        renderer.ResetFrameIndex()
        for _ in range(2):
            renderer.Render(scene, cam)
        assert np.array_equal(bits(renderer.read_accum().reshape(-1, 3)), bits(want["acc"]))
    finally:
        renderer.set_ray_dirs(None)


# ---- mesh through the KD tree ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("order", ORDERS)
@pytest.mark.parametrize("name", ["grid40", "cubesphere16"])
def test_mesh_bit_exact_with_counters(rt, orc, renderer, name, order):
    from raytracer_amd import scenes
    mesh = scenes.displaced_grid(40) if name == "grid40" else scenes.cube_sphere(16)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    got, want = run_both(rt, orc, renderer, scene, osc, 160, 90, 0, 8, 3, stats=True, order=order)
    assert_mesh(got, want, order)
    assert (want["ids"] >= 0).sum() > 500 and (want["ids"] < 0).sum() > 500
    if order == F2B_ORDER:                          # strictly less work (the rays of the direction.x == 0 column, which
        assert got["cnt"].tris_tested < want["cnt"].tris_tested      # nothing can prune, dominate both counts at this size)


@pytest.mark.parametrize("side", [(1.0, 0.3, 0.2), (-0.5, -0.8, 0.6), (0.2, 0.4, -1.0), (-0.7, 0.9, -0.4)])
def test_mesh_orders_agree_from_other_viewpoints(rt, orc, renderer, side):
    """Cameras on different sides of the mesh, looking at it, so that the rays fall into other sign patterns (each pattern
    walks its own node array): both orders against the oracle."""
    from raytracer_amd import scenes
    centre = np.array([0.0, 0.6, -0.8], np.float32)
    scene, osc = make_scene(rt, orc, mesh=scenes.cube_sphere(14, center=tuple(centre), radius=0.7))
    d = np.array(side, np.float32); d /= np.linalg.norm(d)
    cam = rt.Camera(70.0, 0.05, 100.0)
    cam.SetPose((centre + np.float32(1.9) * d).astype(np.float32), (-d).astype(np.float32))
    for order in (REF_ORDER, F2B_ORDER):
        got, want = run_both(rt, orc, renderer, scene, osc, 144, 96, 0, 6, 2, stats=True, cam=cam, order=order)
        assert_mesh(got, want, order)
        # ... and the kernels that are timed: units that are not traced, primary walks from their unit's entry node
        got, want = run_both(rt, orc, renderer, scene, osc, 144, 96, 0, 6, 2, stats=False, cam=cam, order=order)
        assert_exact(got, want)
        assert got["cnt"].entry_nodes_spared > 0
    assert (want["ids"] >= 0).sum() > 1000


@pytest.mark.parametrize("counting", [False, True], ids=["timed-kernels", "counting-kernels"])
@pytest.mark.parametrize("log2_scale", [33, 45, 61])
@pytest.mark.parametrize("order", ORDERS)
def test_scenes_of_absurd_size(rt, orc, renderer, order, log2_scale, counting):
    """The timed front-to-back kernels test boxes with the hardware min / max and divide by the determinant in a short form, which
    is only exact while no operation can overflow: trees within 2^60, triangles with |e1|_1 |e2|_1 <= 2^56.  A scene scaled by 2^33
    (camera included) has triangles beyond the second bound: every ray has to take the reference's own walk instead.  From about
    2^40 the dot products of triIntersect itself overflow: tests succeed at t = +inf, which KDTreeCPU::intersect reports as a hit
    ("intersection_detected") although no distance was accepted, and TraceRay then shades triangle 0 at FLT_MAX; at 2^45 that is a
    part of the pixels, at 2^61 (outside both bounds) every pixel the mesh covers.  Ids, sums and image must be the oracle's, bit
    for bit, in both orders, with zeros planted in direction.x as well (such rays are walked by their wave here: the whole-chip
    kernels are not used where tests can overflow)."""
    from raytracer_amd import scenes
    k = np.float32(2.0 ** log2_scale)
    centre = np.array([0.0, 0.6, -0.8], np.float32)
    verts, tris, normals9, tri_mat, mesh_mats = scenes.cube_sphere(10, center=tuple(centre), radius=0.7)
    scene, osc = make_scene(rt, orc, mesh=((verts * k).astype(np.float32), tris, normals9, tri_mat, mesh_mats), spheres=False)
    d = np.array((0.3, 0.5, 1.0), np.float32); d /= np.linalg.norm(d)
    w, h = 96, 64
    # ray directions of the unscaled camera (the inverse view matrix of a camera 10^10 away loses its rotation to cancellation), origin scaled
    cam0 = rt.Camera(70.0, 0.05, 100.0); cam0.SetPose((centre + np.float32(1.9) * d).astype(np.float32), (-d).astype(np.float32)); cam0.OnResize(w, h)
    dirs = orc.ray_dirs(cam0.GetInverseProjection(), cam0.GetInverseView(), w, h).copy()
    for i, px in enumerate(np.random.default_rng(11).choice(w * h, 40, replace=False)):
        dirs[px, 0] = np.float32(0.0) if i % 2 else np.float32(-0.0)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.SetPose(((centre + np.float32(1.9) * d) * k).astype(np.float32), (-d).astype(np.float32)); cam.OnResize(w, h)
    st = renderer.GetSettings()
    st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed = 0, 4, True, False, True, 1337
    st.TraversalOrder = order
    renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True); renderer.set_stats(counting); renderer.reset_counters()
    renderer.upload(scene, cam)
    renderer.set_ray_dirs(dirs)
    try:
        acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
        cnt = orc.Counters()
        for f in range(1, 3):
            renderer.Render(scene, cam)
            osc.render_frame(orc.Settings(1, 0, 1, 0, 4), cam.GetPosition(), dirs, w, h, f, acc, img, primary_ids=ids if f == 1 else None, counters=cnt)
        got = dict(ids=renderer.read_primary_ids().ravel(), acc=renderer.read_accum().reshape(-1, 3),
                   img=renderer.GetImage().data.reshape(-1, 4), cnt=renderer.read_counters())
        assert_exact(got, dict(ids=ids, acc=acc, img=img, cnt=cnt))
        assert (ids >= 0).sum() > 500
        if log2_scale == 61: assert (ids[ids >= 0] == 0x40000000).all()             # nothing but triangle 0, "detected"
    finally:
        renderer.set_ray_dirs(None); renderer.set_stats(False); st.TraversalOrder = 0


@pytest.mark.parametrize("shift", [0.0, 1e-6, 3e-4], ids=["coincident", "within-rounding", "apart"])
def test_front_to_back_near_ties_fall_back_to_the_reference_walk(rt, orc, renderer, shift):
    """Every triangle twice (the copy under another id and material, lifted by `shift`): each ray that hits the mesh hits two
    different triangles at the same or almost the same distance, and which of them KDTreeCPU::intersect returns depends on the
    order in which it reaches their leaves.  The front-to-back walk has to notice (near_ties counts the rays it walked again in
    the reference's order) and return what the oracle returns, ids and radiance bit for bit.  With near ties on every hit this is
    also the test of the wave-cooperative reference walk: every mesh hit of the image comes out of it."""
    from raytracer_amd import scenes
    verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(24)
    T, V = len(tris), len(verts)
    lifted = verts.copy(); lifted[:, 1] += np.float32(shift)
    red = scenes.make_material((0.9, 0.1, 0.1), (0.0, 0.0, 0.0), roughness=0.7)
    mesh = (np.concatenate([verts, lifted]), np.concatenate([tris, tris + np.uint32(V)]).astype(np.uint32), np.concatenate([normals9, normals9]),
            np.concatenate([tri_mat, np.full(T, len(mesh_mats), np.uint32)]).astype(np.uint32), np.concatenate([mesh_mats, np.array([red], mesh_mats.dtype)]))
    scene, osc = make_scene(rt, orc, mesh=mesh)
    for mode in (0, 2):
        got, want = run_both(rt, orc, renderer, scene, osc, 160, 90, mode, 6, 3, order=F2B_ORDER)
        assert_exact(got, want)
        hits = (want["ids"] >= 0x40000000).sum()
        assert hits > 300
        if shift <= 1e-6:
            assert got["cnt"].near_ties >= hits              # at least every primary ray that hit the mesh


def test_auto_order_follows_the_tree(rt, orc, renderer):
    """A tree of the reference builder over a finely tessellated mesh is a KD tree with complete leaf lists and a small near-tie
    distance: RT_ORDER_AUTO walks it front to back.  The same tree with a child box that no longer reaches the split plane is not a
    KD tree, a tree with a triangle missing from one of its leaves has incomplete lists, and a coarse mesh (large triangles: the
    error bound of the triangle test grows with their size) would repeat too many rays: AUTO takes the reference order for those."""
    from raytracer_amd import scenes, _capi
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(140))
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(64, 48)
    st = renderer.GetSettings(); st.SceneMode, st.TraversalOrder = 0, _capi.ORDER_AUTO
    renderer.OnResize(64, 48); renderer.upload(scene, cam)
    assert renderer.traversal_in_use() == _capi.ORDER_FRONT_TO_BACK
    coarse, _ = make_scene(rt, orc, mesh=scenes.displaced_grid(24))
    renderer.upload(coarse, cam)
    assert renderer.traversal_in_use() == _capi.ORDER_REFERENCE
    a = scene.kd_tree.arrays()
    cnt = None
    for leaf in np.flatnonzero((a["left"] < 0) & (a["right"] < 0) & (a["count"] > 1)):
        tri = a["leaf_idx"][a["first"][leaf] + a["count"][leaf] - 1]
        v = scene.verts[scene.tri_idx[tri]]
        lo, hi = a["center"][leaf] - 0.99 * a["extent"][leaf], a["center"][leaf] + 0.99 * a["extent"][leaf]
        if np.all(v.min(0) < hi) and np.all(v.max(0) > lo):             # well inside the leaf's box, not just touching it
            cnt = a["count"].copy(); cnt[leaf] -= 1                      # the leaf forgets its last triangle
            break
    assert cnt is not None
    scene3 = rt.Scene(); scene3.materials = scene.materials; scene3.spheres = scene.spheres
    scene3.set_triangles(scene.verts, scene.tri_idx, scene.normals9, scene.tri_mat, build_kd=False)
    scene3.kd_tree = rt.FlatKDTree(a["center"], a["extent"], a["left"], a["right"], a["first"], cnt, a["leaf_idx"])
    renderer.upload(scene3, cam)
    assert renderer.traversal_in_use() == _capi.ORDER_REFERENCE
    renderer.upload(scene, cam)
    assert renderer.traversal_in_use() == _capi.ORDER_FRONT_TO_BACK
    a = scene.kd_tree.arrays()
    ext = a["extent"].copy()
    two = np.flatnonzero((a["left"] >= 0) & (a["right"] >= 0))[0]
    ext[a["left"][two]] *= np.float32(0.5)                               # the left child no longer reaches the split plane
    scene2 = rt.Scene(); scene2.materials = scene.materials; scene2.spheres = scene.spheres
    scene2.set_triangles(scene.verts, scene.tri_idx, scene.normals9, scene.tri_mat, build_kd=False)
    scene2.kd_tree = rt.FlatKDTree(a["center"], ext, a["left"], a["right"], a["first"], a["count"], a["leaf_idx"])
    renderer.upload(scene2, cam)
    assert renderer.traversal_in_use() == _capi.ORDER_REFERENCE
    st.TraversalOrder = _capi.ORDER_FRONT_TO_BACK
    assert renderer.traversal_in_use() == _capi.ORDER_FRONT_TO_BACK
    st.TraversalOrder = _capi.ORDER_AUTO


def test_mesh_centred_on_zero_direction_column(rt, orc, renderer):
    """The centre pixel column has direction.x == 0 (1/0 = inf) and the grid is centred on x = 0, so (center.x - origin.x)
    * inf = NaN in the slab test: the reference counts such boxes as hit and never prunes them."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(32))
    for order in (REF_ORDER, F2B_ORDER):
        got, want = run_both(rt, orc, renderer, scene, osc, 64, 48, 0, 4, 2, stats=True, order=order)
        assert_mesh(got, want, order)
    col = want["ids"].reshape(48, 64)[:, 32]
    assert (col >= 0).any()


def test_reference_built_tree_and_reference_hits(rt, orc, renderer):
    """Upload the tree the REFERENCE's KDTreeCPU built (golden fixture) and compare the primary ids with the ids the
    reference's own KDTreeCPU::intersect returned."""
    from raytracer_amd import scenes
    g = np.load(os.path.join(GOLD, "ref_kd_grid40.npz"))
    verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(40)
    scene = rt.Scene()
    scene.materials = mesh_mats
    scene.set_triangles(verts, tris, normals9, tri_mat, build_kd=False)
    scene.kd_tree = rt.FlatKDTree(g["center"], g["extent"], g["left"], g["right"], g["first"], g["count"], g["leaf_idx"])
    w, h = (int(x) for x in g["width"])
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings(); st.SceneMode, st.Bounces, st.AntiAliasing = 0, 1, False
    renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True)
    renderer.Render(scene, cam)
    assert np.array_equal(renderer.read_primary_ids().ravel(), g["primary_ids"])


@pytest.mark.parametrize("order", ORDERS)
def test_spheres_and_mesh_together(rt, orc, renderer, order):
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.cube_sphere(12, center=(0.2, 0.7, -0.6), radius=0.5))
    got, want = run_both(rt, orc, renderer, scene, osc, 192, 108, 2, 6, 3, stats=True, order=order)
    assert_mesh(got, want, order, extra=("sphere_tests",))
    tri = (want["ids"] >= 0x40000000).sum(); sph = ((want["ids"] >= 0) & (want["ids"] < 0x40000000)).sum()
    assert tri > 300 and sph > 300


# ---- Renderer semantics ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [1, 0, 2], ids=["spheres", "mesh", "spheres+mesh"])
def test_frames_in_one_launch_equal_separate_launches(rt, orc, renderer, mode):
    """Several frames per launch (a lane keeps its pixel and adds the frames in order) against one launch per frame and the
    oracle; in the tree modes the paths also change waves on the way (most waves of so small a frame start without work)."""
    from raytracer_amd import scenes
    mesh = None if mode == 1 else scenes.displaced_grid(40, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    got, want = run_both(rt, orc, renderer, scene, osc, 160, 96, mode, 5, 6)
    assert_exact(got, want)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(160, 96)
    renderer.ResetFrameIndex()
    renderer.Render(scene, cam, frames=4)
    renderer.Render(scene, cam, frames=2)
    assert renderer.GetFrameIndex() == 7
    assert np.array_equal(bits(renderer.read_accum().reshape(-1, 3)), bits(want["acc"]))
    assert np.array_equal(bits(renderer.GetImage().data.reshape(-1, 4)), bits(want["img"]))


@pytest.mark.parametrize("mode", [1, 0, 2], ids=["spheres", "mesh", "spheres+mesh"])
def test_sixteen_frames_per_launch_and_deep_accumulation(rt, orc, renderer, mode):
    """Launches of 16 Renders (tree scenes: every (pixel, frame) is a path of its own whose radiance goes to that frame's slot,
    the resolve pass adds the slots in frame order) against the oracle's frame-by-frame sum: 37 frames as 16 + 16 + 5 (one call,
    split by the library in tree scenes), bit-exact, then on to 133 frames -- accumulation deeper than any other test."""
    from raytracer_amd import scenes
    w, h = 128, 72
    mesh = None if mode == 1 else scenes.displaced_grid(32, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings()
    st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed, st.TraversalOrder = mode, 5, True, False, True, 1337, 0
    renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True); renderer.set_stats(False); renderer.reset_counters()
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
    ost = orc.Settings(1, mode, 1, 0, 5)
    cnt = orc.Counters()
    for f in range(1, 38):
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img, primary_ids=ids if f == 1 else None, counters=cnt)
    renderer.Render(scene, cam, frames=37)
    assert renderer.GetFrameIndex() == 38
    got = dict(ids=renderer.read_primary_ids().ravel(), acc=renderer.read_accum().reshape(-1, 3), img=renderer.GetImage().data.reshape(-1, 4),
               cnt=renderer.read_counters())
    assert_exact(got, dict(ids=ids, acc=acc, img=img, cnt=cnt))
    for f in range(38, 134):
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img)
    renderer.Render(scene, cam, frames=16)
    renderer.Render(scene, cam, frames=80)
    assert renderer.GetFrameIndex() == 134
    assert np.array_equal(bits(renderer.read_accum().reshape(-1, 3)), bits(acc))
    assert np.array_equal(bits(renderer.GetImage().data.reshape(-1, 4)), bits(img))


@pytest.mark.parametrize("mode", [1, 2], ids=["spheres", "spheres+mesh"])
def test_thousand_frame_accumulation(rt, orc, renderer, mode):
    """1024 accumulated frames (the depth of the progressive configuration) at a small size, as launches of up to 16 Renders,
    against the oracle's frame-by-frame sum: every bit of the sum and of the last frame's image."""
    from raytracer_amd import scenes
    w, h, n = 48, 32, 1024
    mesh = None if mode == 1 else scenes.displaced_grid(16, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings()
    st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed, st.TraversalOrder = mode, 5, True, False, True, 1337, 0
    renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.set_stats(False)
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    ost = orc.Settings(1, mode, 1, 0, 5)
    for f in range(1, n + 1):
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img)
    renderer.Render(scene, cam, frames=1000)
    renderer.Render(scene, cam, frames=24)
    assert renderer.GetFrameIndex() == n + 1
    assert np.array_equal(bits(renderer.read_accum().reshape(-1, 3)), bits(acc))
    assert np.array_equal(bits(renderer.GetImage().data.reshape(-1, 4)), bits(img))


def test_accumulate_off_and_reset(rt, orc, renderer):
    scene, osc = make_scene(rt, orc)
    w, h = 100, 60
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings(); st.SceneMode, st.Bounces, st.Accumulate, st.AntiAliasing, st.UseACE_Color = 1, 5, False, False, True
    renderer.OnResize(w, h); renderer.ResetFrameIndex()
    renderer.Render(scene, cam); a1 = renderer.read_accum().copy()
    renderer.Render(scene, cam); a2 = renderer.read_accum().copy()
    assert renderer.GetFrameIndex() == 1 and np.array_equal(bits(a1), bits(a2))      # frame 1 re-rendered, not summed
    st.Accumulate = True
    renderer.Render(scene, cam); renderer.Render(scene, cam)
    assert renderer.GetFrameIndex() == 3
    renderer.ResetFrameIndex()                                                       # what a camera move does
    renderer.Render(scene, cam)
    assert np.array_equal(bits(renderer.read_accum()), bits(a1))
    renderer.OnResize(w, h)                                                          # unchanged size: no reset
    assert renderer.GetFrameIndex() == 2


def test_live_edits_between_frames_keep_accumulating(rt, orc, renderer):
    """The UI edits spheres/materials between frames without resetting accumulation (RaytracerApp.cpp:283-331)."""
    scene, osc = make_scene(rt, orc)
    w, h = 120, 72
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings(); st.SceneMode, st.Bounces, st.Accumulate, st.AntiAliasing, st.UseACE_Color, st.Seed = 1, 5, True, False, True, 1337
    renderer.OnResize(w, h); renderer.ResetFrameIndex()
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    ost = orc.Settings(1, 1, 1, 0, 5)
    for f in range(1, 4):
        if f == 2:
            scene.spheres = scene.spheres.copy(); scene.spheres[5]["y"] = 0.6; scene.spheres[7]["radius"] = 0.35
            scene.materials = scene.materials.copy(); scene.materials[1]["albedo"] = (0.1, 0.2, 0.9)
            osc.set_spheres(scene.spheres); osc.set_materials(scene.materials)
        renderer.Render(scene, cam)
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img)
    assert np.array_equal(bits(renderer.read_accum().reshape(-1, 3)), bits(acc))


def test_errors_are_reported_not_faulted(rt, orc, renderer):
    from raytracer_amd import _capi
    r2 = rt.Renderer(0)
    scene, _ = make_scene(rt, orc)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(32, 32)
    with pytest.raises(_capi.RtError):
        r2.render_only(1)                                            # before OnResize
    r2.OnResize(32, 32)
    with pytest.raises(_capi.RtError):
        r2.render_only(1)                                            # before a camera
    r2.GetSettings().SceneMode = 0
    with pytest.raises(_capi.RtError):
        r2.Render(scene, cam)                                        # mesh mode without a tree
    r2.GetSettings().SceneMode = 1
    bad = rt.Scene(); bad.spheres = scene.spheres.copy(); bad.spheres[2]["material"] = 700; bad.materials = scene.materials
    with pytest.raises(_capi.RtError):
        r2.Render(bad, cam)                                          # material index out of range
    with pytest.raises(_capi.RtError):
        r2.render_only(0)                                            # frame indices are 1-based
    r2.Render(scene, cam)                                            # and the context still works afterwards


# ---- partitions for several GPUs, exercised on one ----------------------------------------------------------------------
def test_tile_split_bands_assemble_the_frame(rt, orc, renderer):
    from raytracer_amd import dist as rtd
    scene, osc = make_scene(rt, orc)
    w, h = 200, 150
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, 1, 5, 2)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    world = 3
    out = np.zeros((h, w, 3), np.float32)
    for rank in range(world):
        rr = rt.Renderer(0)
        s = rr.GetSettings(); s.SceneMode, s.Bounces = 1, 5
        rr.OnResize(w, h); rr.set_bands(rank, world)
        rr.Render(scene, cam); rr.Render(scene, cam)
        rows = rtd.band_rows(h, rank, world)
        out[rows] = rr.read_accum()[rows]
    assert np.array_equal(bits(out.reshape(-1, 3)), bits(want["acc"]))


@pytest.mark.parametrize("mode", [1, 2], ids=["spheres", "spheres+mesh"])
def test_tile_split_with_the_3x3_filter(rt, orc, renderer, mode):
    """The filter (Renderer.cpp:84-124) reads the neighbours' sums, so under a tile split a context also traces and sums the rows
    bordering its bands: three band contexts with the filter on, three frames (the last two in one launch), assembled image and
    accumulation against the oracle's filtered frame, bit-exact.  Switching the filter on in the middle of an accumulation is
    refused until the accumulation restarts."""
    from raytracer_amd import dist as rtd, scenes, _capi
    mesh = None if mode == 1 else scenes.displaced_grid(24, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    w, h = 150, 100
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, mode, 5, 3, aa=True)
    assert_exact(got, want)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    world = 3
    img = np.zeros((h, w, 4), np.float32); acc = np.zeros((h, w, 3), np.float32)
    for rank in range(world):
        rr = rt.Renderer(0)
        s = rr.GetSettings(); s.SceneMode, s.Bounces, s.AntiAliasing = mode, 5, True
        rr.OnResize(w, h); rr.set_bands(rank, world)
        rr.Render(scene, cam); rr.Render(scene, cam, frames=2)
        rows = rtd.band_rows(h, rank, world)
        img[rows] = rr.GetImage().data[rows]; acc[rows] = rr.read_accum()[rows]
        if rank == 1:
            s.AntiAliasing = False; rr.Render(scene, cam)            # a frame without the bordering rows ...
            s.AntiAliasing = True
            with pytest.raises(_capi.RtError):
                rr.Render(scene, cam)                                # ... and the filter may not read them any more
            rr.ResetFrameIndex(); rr.Render(scene, cam)              # until the accumulation restarts
    assert np.array_equal(bits(acc.reshape(-1, 3)), bits(want["acc"]))
    assert np.array_equal(bits(img.reshape(-1, 4)), bits(want["img"]))


def test_units_without_a_traced_path_through_live_changes(rt, orc):
    """Mesh-only scene, most work units without a primary ray into the tree.  After the first Render of an accumulation the resolve
    pass visits only the listed units (the others hold a sum of +0, its image, ids of -1).  Whatever changes in the middle of the
    accumulation -- the tone map, the id recording, the camera (units that had radiance become units without a path), the 3x3
    filter on and off, several Renders per launch -- the frame, the sums and the ids stay the oracle's, bit for bit."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(24, x_range=(-0.8, 0.8), z_range=(-1.0, 1.0), mat_offset=0), spheres=False)
    w, h = 160, 96
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    r = rt.Renderer(0)
    st = r.GetSettings(); st.SceneMode, st.Bounces, st.UseACE_Color = 0, 4, True
    r.OnResize(w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    frame = [1]

    def step(frames=1, use_ace=True, aa=False):
        st.UseACE_Color, st.AntiAliasing = use_ace, aa
        r.Render(scene, cam, frames=frames)
        dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
        ost = orc.Settings(1, 0, int(use_ace), int(aa), 4)
        for _ in range(frames):
            osc.render_frame(ost, cam.GetPosition(), dirs, w, h, frame[0], acc, img, seed=st.Seed)
            frame[0] += 1
        assert np.array_equal(bits(r.read_accum().reshape(-1, 3)), bits(acc)), "sums at frame %d" % (frame[0] - 1)
        assert np.array_equal(bits(r.GetImage().data.reshape(-1, 4)), bits(img)), "image at frame %d" % (frame[0] - 1)

    def oracle_ids():
        ids = np.zeros(h * w, np.int32); a = np.zeros((h * w, 3), np.float32); i = np.zeros((h * w, 4), np.float32)
        dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
        osc.render_frame(orc.Settings(1, 0, 1, 0, 4), cam.GetPosition(), dirs, w, h, 1, a, i, seed=st.Seed, primary_ids=ids)
        return ids

    step(); step(); step()
    first = oracle_ids()
    assert (first < 0).mean() > 0.5 and (first >= 0).mean() > 0.02          # most of the frame misses the mesh, some of it hits
    step(use_ace=False); step(use_ace=True)                    # the tone map changes between Renders
    r.record_ids(True); step()
    assert np.array_equal(r.read_primary_ids().ravel(), oracle_ids())
    cam.SetPose(np.array([0.9, 1.0, 3.0], np.float32), np.array([0.0, -0.1, -1.0], np.float32) / np.float32(np.sqrt(1.01)))
    cam.OnResize(w, h)
    step(); step(frames=3)                                     # the camera moved, the accumulation goes on
    assert np.array_equal(r.read_primary_ids().ravel(), oracle_ids())
    step(aa=True); step(aa=False); step(frames=2)              # the filter on for one Render
    r.ResetFrameIndex(); acc[:] = 0; frame[0] = 1              # a new accumulation with the moved camera
    step(); step(frames=4); step()
    assert np.array_equal(r.read_primary_ids().ravel(), oracle_ids())


def test_kernel_times_are_taken_only_when_asked_for(rt, orc):
    """A context records no events around its launches until it is asked for kernel times; from then on around every launch, or
    around launch n / 2 of every n (rt_set_kernel_timing), counted from the last reset.  The frames do not depend on it."""
    scene, osc = make_scene(rt, orc)
    w, h = 96, 64
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    r = rt.Renderer(0)
    st = r.GetSettings(); st.SceneMode, st.Bounces = 1, 3
    r.OnResize(w, h)
    for _ in range(5): r.Render(scene, cam)
    assert r.kernel_time().launches == 0                     # nobody had asked
    for _ in range(3): r.Render(scene, cam)
    kt = r.kernel_time(reset=True)
    assert kt.launches == 3 and kt.frames == 3 and kt.total_ms > 0.0
    r.set_kernel_timing(4)
    for _ in range(10): r.Render(scene, cam)                 # launches 2 and 6 of 0..9
    assert r.kernel_time(reset=True).launches == 2
    r.Render(scene, cam, frames=3); r.Render(scene, cam, frames=3); r.Render(scene, cam, frames=3)     # launch 2 of 0..2, three Renders
    kt = r.kernel_time()
    assert kt.launches == 1 and kt.frames == 3
    r.set_kernel_timing(0)
    for _ in range(4): r.Render(scene, cam)
    assert r.kernel_time().launches == 1
    img = r.GetImage().data.copy()
    r2 = rt.Renderer(0)
    s2 = r2.GetSettings(); s2.SceneMode, s2.Bounces = 1, 3
    r2.OnResize(w, h)
    for _ in range(5 + 3 + 10 + 9 + 4): r2.Render(scene, cam)
    r2.Render(scene, cam, frames=2)
    r.Render(scene, cam, frames=2)
    assert np.array_equal(bits(r.GetImage().data.reshape(-1, 4)), bits(r2.GetImage().data.reshape(-1, 4)))
    assert img.shape == (h, w, 4)


def test_pipelined_hand_off_returns_the_previous_frame(rt, orc, renderer):
    """rt_map_image_pipelined: the copy of frame k runs beside the trace of frame k + 1; each call returns the frame of the call
    before it (the first its own).  Every returned frame is bit-exact."""
    scene, osc = make_scene(rt, orc)
    w, h, n = 200, 120, 5
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    ost = orc.Settings(1, 1, 1, 0, 5)
    frames = []
    for f in range(1, n + 1):
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img)
        frames.append(img.copy())
    st = renderer.GetSettings(); st.SceneMode, st.Bounces, st.AntiAliasing, st.UseACE_Color, st.Accumulate, st.Seed = 1, 5, False, True, True, 1337
    renderer.OnResize(w, h); renderer.ResetFrameIndex()
    renderer.pipelined = True
    try:
        for f in range(1, n + 1):
            renderer.Render(scene, cam)
            shown = max(f - 1, 1)
            assert np.array_equal(bits(renderer.GetImage().data.reshape(-1, 4)), bits(frames[shown - 1])), "call %d" % f
            assert np.array_equal(bits(renderer.read_image().reshape(-1, 4)), bits(frames[f - 1]))       # the newest frame, on request
    finally:
        renderer.pipelined = False
    assert np.array_equal(bits(renderer.map_image().reshape(-1, 4)), bits(frames[-1]))


def test_pipelined_hand_off_of_a_mesh_scene_with_units_that_are_not_traced(rt, orc):
    """The pipelined hand-off changes frame buffers at every call, and in a mesh-only scene the resolve pass of the later Renders
    of an accumulation leaves the units without a traced path alone.  Each of the two buffers must have been resolved in full
    before that: every frame shown -- through four accumulated frames, a camera move, a restart -- is the oracle's, bit for bit."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(24, x_range=(-0.8, 0.8), z_range=(-1.0, 1.0), mat_offset=0), spheres=False)
    w, h = 160, 96
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    r = rt.Renderer(0)
    st = r.GetSettings(); st.SceneMode, st.Bounces, st.UseACE_Color = 0, 4, True
    r.OnResize(w, h)
    r.pipelined = True
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    shown = []                                              # the oracle's frames in the order they were rendered
    frame = 1

    def step():
        nonlocal frame
        dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
        osc.render_frame(orc.Settings(1, 0, int(st.UseACE_Color), 0, 4), cam.GetPosition(), dirs, w, h, frame, acc, img, seed=st.Seed)
        frame += 1
        shown.append(img.copy())
        r.Render(scene, cam)
        want = shown[-2] if len(shown) > 1 else shown[-1]  # each call returns the frame of the call before it (the first its own)
        assert np.array_equal(bits(r.GetImage().data.reshape(-1, 4)), bits(want)), "call %d" % len(shown)
        assert np.array_equal(bits(r.read_image().reshape(-1, 4)), bits(shown[-1])), "newest frame at call %d" % len(shown)

    for _ in range(5): step()
    assert (shown[-1][:, :3].sum(axis=1) == 0).mean() > 0.5          # most of the frame belongs to units without a path
    cam.SetPose(np.array([0.9, 1.0, 3.0], np.float32), np.array([0.0, -0.1, -1.0], np.float32) / np.float32(np.sqrt(1.01)))
    cam.OnResize(w, h)
    for _ in range(4): step()                              # the camera moved, the accumulation goes on (other units are empty now)
    r.ResetFrameIndex(); acc[:] = 0; frame = 1
    for _ in range(5): step()                              # a new accumulation
    st.UseACE_Color = False
    r.ResetFrameIndex(); acc[:] = 0; frame = 1
    for _ in range(3): step()                              # ... and one without the tone map
    assert np.array_equal(bits(r.read_accum().reshape(-1, 3)), bits(acc))


def test_minus_zero_x_direction_primaries_in_a_mesh_only_scene(rt, orc, renderer):
    """Primary rays with direction.x == -0 are no column rays (those have +0): in a mesh-only scene their wave walks the tree for
    them, and they take no part in finding the node at which their unit's other primary walks start.  Host-supplied directions
    with -0 and +0 planted in x across whole units and in single pixels; ids, sums and image are the oracle's in both orders."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(40, mat_offset=0), spheres=False)
    w, h = 96, 64
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h).reshape(h, w, 3).copy()
    dirs[10:40, 20, 0] = -0.0                               # a column of -0 through many units
    dirs[30, 40:60, 0] = -0.0                               # a row: whole units of -0
    dirs[34, 41:57:3, 0] = 0.0                              # +0 and -0 mixed with ordinary rays inside units
    dirs[35, 42:58:3, 0] = -0.0
    dirs = dirs.reshape(-1, 3)
    for order in (REF_ORDER, F2B_ORDER):
        st = renderer.GetSettings()
        st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed, st.TraversalOrder = 0, 4, True, False, True, 1337, order
        renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True)
        renderer.upload(scene, cam); renderer.set_ray_dirs(dirs)
        acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
        try:
            for f in range(1, 4):
                renderer.Render(scene, cam)
                osc.render_frame(orc.Settings(1, 0, 1, 0, 4), cam.GetPosition(), dirs, w, h, f, acc, img, primary_ids=ids if f == 1 else None)
            got = dict(ids=renderer.read_primary_ids().ravel(), acc=renderer.read_accum().reshape(-1, 3),
                       img=renderer.GetImage().data.reshape(-1, 4), cnt=None)
        finally:
            renderer.set_ray_dirs(None); st.TraversalOrder = 0
        assert_exact(got, dict(ids=ids, acc=acc, img=img, cnt=None), counters=())
        assert (ids[(np.arange(h * w) % w == 20) & (np.arange(h * w) // w >= 10) & (np.arange(h * w) // w < 40)] >= 0).any()


def test_sample_split_sum_matches_within_reassociation_tolerance(rt, orc, renderer):
    from raytracer_amd import dist as rtd
    scene, osc = make_scene(rt, orc)
    w, h, world, per = 160, 100, 4, 3
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, 1, 5, world * per)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    total = np.zeros((h, w, 3), np.float32)
    for rank in range(world):
        rr = rt.Renderer(0)
        s = rr.GetSettings(); s.SceneMode, s.Bounces = 1, 5
        rr.OnResize(w, h); rr.upload(scene, cam); rr.reset_accum()
        for f in rtd.frames_for_rank(rank, world, per):
            rr.render_only(f)
        total += rr.read_accum()
    ref = want["acc"].reshape(h, w, 3)
    assert np.all(np.abs(total - ref) <= 1e-4 + 1e-4 * np.abs(ref))


# ---- full-size configurations ------------------------------------------------------------------------------------------------
def test_config2_1080p_sphere_scene_ids_and_accumulation(rt, orc, renderer):
    """BASELINE configs[1] at full size: all 2,073,600 primary ids and two frames of accumulation, bit-exact."""
    scene, osc = make_scene(rt, orc)
    got, want = run_both(rt, orc, renderer, scene, osc, 1920, 1080, 1, 5, 2, threads=16)
    assert_exact(got, want)
    # all 64 frames of the configuration: deterministic run to run, and every 16th row equal to the oracle's 64-frame sum
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(1920, 1080)
    sums = []
    renderer.download = False
    try:
        for _ in range(2):
            renderer.ResetFrameIndex()
            for _f in range(64):
                renderer.Render(scene, cam)
            sums.append(renderer.read_accum())
    finally:
        renderer.download = True
    assert renderer.GetFrameIndex() == 65
    assert np.array_equal(bits(sums[0]), bits(sums[1]))
    assert np.isfinite(sums[0]).all() and (sums[0].sum(axis=2) > 0).mean() > 0.5
    w, h = 1920, 1080
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    ost = orc.Settings(1, 1, 1, 0, 5)
    for f in range(1, 65):
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img, nthreads=16, row_step=16)
    rows = np.arange(0, h, 16)
    assert np.array_equal(bits(sums[0][rows]), bits(acc.reshape(h, w, 3)[rows]))


@pytest.mark.parametrize("order", ORDERS)
def test_cornell_class_obj_through_the_ingest(rt, orc, renderer, order, tmp_path):
    """The reference application's own kind of scene: a Cornell-box-class OBJ + MTL (wall-sized triangles, a light, two blocks and a
    water surface in the material the application makes transparent) through the ingest, the application's materials appended;
    mesh-only, 8 bounces.  RT_ORDER_AUTO walks such a mesh in the reference's order (the near-tie distance grows with the square
    of the largest triangle); both orders give the oracle's ids, sums and image."""
    from raytracer_amd import ingest, scenes
    verts, tris, normals9, tri_mat, mats, names = ingest.load_obj(scenes.write_cornell_class_obj(str(tmp_path), 16), app_materials=True)
    assert "water" in names and mats[names.index("water")]["transparency"] == 1.0
    scene = rt.Scene(); scene.materials = mats; scene.spheres = scenes.default_spheres(mat_offset=len(names))
    scene.set_triangles(verts, tris, normals9, tri_mat, max_nodes=1_000_000, nthreads=4)
    osc = orc.Scene(); osc.set_spheres(scene.spheres); osc.set_materials(mats); osc.set_mesh(verts, tris, normals9, tri_mat)
    osc.import_kd(scene.kd_tree.arrays())
    got, want = run_both(rt, orc, renderer, scene, osc, 200, 120, 0, 8, 4, stats=True, order=order)
    if order == REF_ORDER:
        assert_mesh(got, want, order)
    else:
        # exact, but no saving here: with wall-sized triangles the near-tie distance is of the order of the hit distance itself, so the
        # front-to-back walk prunes almost nothing by it and repeats most rays that hit twice in the reference's order
        assert_exact(got, want, ("rays",))
        assert got["cnt"].near_ties > 0
    assert (want["ids"] >= 0).mean() > 0.3 and want["acc"].sum() > 0             # the box fills a good part of the view and its light reaches it
    st = renderer.GetSettings(); st.TraversalOrder = 0
    renderer.upload(scene, rt.Camera(70.0, 0.05, 100.0))
    assert renderer.traversal_in_use() == REF_ORDER


def test_config1_720p_sphere_scene_hundred_accumulated_frames(rt, orc, renderer):
    """BASELINE configs[0] at its own size and depth: the default sphere scene at 1280x720, 5 bounces, 100 accumulated frames (the
    reference's CPU-runnable case).  The oracle renders every 8th row of every frame (rows are independent: a pixel's random
    stream is its own); on those rows the sums after 100 frames, the 100th frame's image and the ids are the library's bits."""
    scene, osc = make_scene(rt, orc)
    w, h, frames, step = 1280, 720, 100, 8
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings()
    st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed = 1, 5, True, False, True, 1337
    renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True); renderer.reset_counters()
    renderer.download = False
    try:
        for _ in range(frames):
            renderer.Render(scene, cam)                        # one launch per Render, as the application makes them
    finally:
        renderer.download = True
    assert renderer.GetFrameIndex() == frames + 1
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.full(h * w, -7, np.int32)
    cnt = orc.Counters()
    ost = orc.Settings(1, 1, 1, 0, 5)
    for f in range(1, frames + 1):
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img, primary_ids=ids if f == 1 else None, counters=cnt,
                         nthreads=16, row_step=step)
    rows = np.arange(0, h, step)
    got_acc = renderer.read_accum().reshape(h, w, 3)[rows]; got_img = renderer.read_image().reshape(h, w, 4)[rows]
    got_ids = renderer.read_primary_ids().reshape(h, w)[rows]
    assert np.array_equal(got_ids, ids.reshape(h, w)[rows])
    assert np.array_equal(bits(got_acc), bits(acc.reshape(h, w, 3)[rows])), "sums after %d frames" % frames
    assert np.all(np.abs(got_acc - acc.reshape(h, w, 3)[rows]) <= 1e-5 + 1e-5 * np.abs(acc.reshape(h, w, 3)[rows]))     # the stated tolerance
    assert np.array_equal(bits(got_img), bits(img.reshape(h, w, 4)[rows])), "image of frame %d" % frames
    # the rays of the rows the oracle rendered, scaled: the library traced every row (3.668 rays per pixel-sample in this scene)
    rays = renderer.read_counters().rays
    assert abs(rays / (w * h * frames) - cnt.rays / (w * len(rows) * frames)) < 0.02


def test_config3_70k_mesh_1080p_primary_ids(rt, orc, renderer):
    """BASELINE configs[2] at full size: primary ids of all 1080p pixels through the 69,938-triangle tree, bit-exact;
    then one 8-bounce frame at a sixteenth of the rows against the oracle."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(187), spheres=False)
    w, h = 1920, 1080
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings(); st.SceneMode, st.Bounces, st.AntiAliasing, st.Accumulate, st.UseACE_Color, st.Seed = 0, 8, False, True, True, 1337
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    ids, *_ = osc.trace_primary(0, cam.GetPosition(), dirs)
    assert (ids >= 0).sum() > 100000
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    osc.render_frame(orc.Settings(1, 0, 1, 0, 8), cam.GetPosition(), dirs, w, h, 1, acc, img, nthreads=16, row_step=16)
    rows = np.arange(0, h, 16)
    full = {}
    try:
        for order in (REF_ORDER, F2B_ORDER):
            st.TraversalOrder = order
            renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True)
            renderer.Render(scene, cam)
            assert np.array_equal(renderer.read_primary_ids().ravel(), ids)
            full[order] = renderer.read_accum()
            assert np.array_equal(bits(full[order][rows]), bits(acc.reshape(h, w, 3)[rows]))
    finally:
        st.TraversalOrder = 0
    assert np.array_equal(bits(full[REF_ORDER]), bits(full[F2B_ORDER]))      # every pixel, not only the oracle's rows


def test_configs4_and_5_ten_million_triangles_4k(rt, orc, renderer):
    """BASELINE configs[3] and [4] at full size: the 9,999,392-triangle mesh (KD tree of ~16 M nodes, ~7 GB in HBM) at
    3840x2160, mesh alone (8 bounces) and with the default spheres.  Against the oracle: every 16th row of two accumulated frames
    (the oracle needs ~40 s per full frame on 16 cores), ids and accumulation bit-exact in both walk orders, which also covers
    the column of direction.x == 0.  Over all pixels the two orders are compared with each other (see below)."""
    import bench
    cfg = bench.CONFIGS[4]
    scene = bench.build_scene(cfg)
    osc = orc.Scene()
    osc.set_spheres(scene.spheres); osc.set_materials(scene.materials)
    osc.set_mesh(scene.verts, scene.tri_idx, scene.normals9, scene.tri_mat)
    osc.import_kd(scene.kd_tree.arrays())
    w, h = cfg["w"], cfg["h"]
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    rows = np.arange(0, h, 16)
    st = renderer.GetSettings()
    renderer.download = False
    try:
        for mode, bounces in ((0, 8), (2, 8)):
            st.SceneMode, st.Bounces, st.AntiAliasing, st.Accumulate, st.UseACE_Color, st.Seed = mode, bounces, False, True, True, 1337
            acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.full(h * w, -7, np.int32)
            for f in (1, 2):
                osc.render_frame(orc.Settings(1, mode, 1, 0, bounces), cam.GetPosition(), dirs, w, h, f, acc, img,
                                 primary_ids=ids if f == 1 else None, nthreads=16, row_step=16)
            full = {}
            for order in (REF_ORDER, F2B_ORDER):
                st.TraversalOrder = order
                renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True)
                renderer.Render(scene, cam, frames=2)
                full[order] = (renderer.read_primary_ids().copy(), renderer.read_accum().copy())
                assert np.array_equal(full[order][0][rows], ids.reshape(h, w)[rows]), "mode %d order %d: primary ids" % (mode, order)
                assert np.array_equal(bits(full[order][1][rows]), bits(acc.reshape(h, w, 3)[rows])), "mode %d order %d: accumulation" % (mode, order)
            # all 8.3 M pixels: the front-to-back walk repeats, in the reference's order, every ray that hit two different triangles
            # within rounding distance of each other (before it did, one primary id and one radiance pixel of mode 0 differed)
            differ = np.flatnonzero(full[REF_ORDER][0].ravel() != full[F2B_ORDER][0].ravel())
            assert len(differ) == 0, "%d primary ids differ between the walk orders" % len(differ)
            acc_differ = (bits(full[REF_ORDER][1]) != bits(full[F2B_ORDER][1])).any(axis=-1).sum()
            assert acc_differ == 0, "%d pixels of accumulated radiance differ between the walk orders" % acc_differ
            print("mode %d: %d primary ids, %d radiance pixels differ between walk orders (of %d)" % (mode, len(differ), acc_differ, w * h))
            # kept first hits at full size (auto order = front to back here): the same ids and sums again
            st.TraversalOrder, st.ReuseFirstHits = 0, True
            renderer.ResetFrameIndex(); renderer.Render(scene, cam, frames=2)
            st.ReuseFirstHits = False
            assert np.array_equal(renderer.read_primary_ids(), full[F2B_ORDER][0]), "mode %d: primary ids with kept first hits" % mode
            assert np.array_equal(bits(renderer.read_accum()), bits(full[F2B_ORDER][1])), "mode %d: accumulation with kept first hits" % mode
            hit = full[F2B_ORDER][0] >= 0
            assert hit.mean() > 0.15 and (full[F2B_ORDER][0] >= 0x40000000).sum() > 200000   # the mesh is seen
    finally:
        st.TraversalOrder = 0; st.ReuseFirstHits = False; renderer.download = True


# ---- plumbing for the multi-GPU path, on one GPU ------------------------------------------------------------------------------
def test_accum_tensor_aliases_the_device_buffer(rt, orc, renderer):
    """raytracer_amd.dist hands torch a zero-copy view of the accumulation buffer (what the RCCL sum runs on)."""
    import torch
    from raytracer_amd import dist as rtd
    scene, osc = make_scene(rt, orc)
    got, want = run_both(rt, orc, renderer, scene, osc, 64, 48, 1, 5, 2)
    t = rtd.accum_tensor(renderer)
    assert t.shape == (48, 64, 4) and t.is_cuda
    assert np.array_equal(bits(t[..., :3].cpu().numpy().reshape(-1, 3)), bits(want["acc"]))
    t.mul_(2.0)                                   # written through torch ...
    torch.cuda.synchronize()
    assert np.array_equal(bits(renderer.read_accum().reshape(-1, 3)), bits(want["acc"] * np.float32(2.0)))   # ... seen by the library
    renderer.resolve(2)
    img = renderer.read_image().reshape(-1, 4)
    ref = np.zeros(3, np.float32); out = np.zeros(3, np.float32)
    import ctypes as C
    for i in (0, 777, 3071):
        ref[:] = want["acc"][i] * np.float32(2.0) / np.float32(2.0)
        orc.lib().orc_tonemap(ref.ctypes.data_as(C.c_void_p), 1, 1, out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(bits(img[i, :3]), bits(out))


def test_bench_distributed_code_path_with_one_rank():
    """bench.py under torch.distributed.run with RCCL (one rank, RT_BENCH_FORCE_DIST=1): process group, sum of the
    accumulation buffers, rank-0 resolve and the JSON line."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--config", "1",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["config"]["split"] == "sample"
    assert d["roofline"]["frac"] > 0 and d["unit"] == "Mrays/s"


def test_bench_with_two_ranks_sharing_the_gpu():
    """bench.py --gpus 2 as the driver launches it, both ranks on device 0 and the collectives over gloo (RCCL refuses two ranks on one
    GPU): every rank's code path -- its frame indices, the sum onto rank 0, the maximum of the times, the one JSON line -- runs; the
    rays of the line are those of both ranks' frames."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--config", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["config"]["split"] == "sample" and d["scaling"] == "weak" and "rehearsal" in d
    # 8 frames of the 1280x720 sphere scene: 3.38 M rays each
    assert 7.5 * 3.3e6 < d["value"] * d["ms_per_step"] * 1e3 * 4 < 8.5 * 3.5e6
    assert "cpu_baseline" not in d                           # the oracle is timed beside the one-GPU run only


# ---- the drop-in itself: RendererHIP with the reference's own Scene / Camera / KDTreeCPU ------------------------------------------
@pytest.mark.parametrize("use_spheres,contexts", [(1, 1), (0, 1), (0, 3)], ids=["spheres", "mesh", "mesh-3-contexts"])
def test_renderer_hip_with_reference_host_types(orc, tmp_path, use_spheres, contexts):
    """oracle/_ref/renderer_hip_demo links the reference's unmodified Scene.h, Camera.cpp and KDTreeCPU (its real builder)
    with include/RendererHIP.h in place of Renderer; what it hands to Image::SetData and its accumulation buffer must equal
    the oracle's.  (The binary is built where the reference tree exists and travels with the repo.)"""
    import subprocess
    from raytracer_amd import scenes
    demo = os.path.join(ROOT, "oracle", "_ref", "renderer_hip_demo")
    if not os.path.exists(demo):
        pytest.skip("oracle/_ref/renderer_hip_demo not built (reference tree absent at build time)")
    w, h, frames, bounces = 96, 64, 3, 5 if use_spheres else 4
    spheres, mats = scenes.default_spheres(), scenes.default_materials()
    verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(20, mat_offset=len(mats))
    mats = np.concatenate([mats, mesh_mats])
    with open(tmp_path / "scene.bin", "wb") as f:
        f.write(np.array([w, h, frames, bounces, use_spheres, len(spheres), len(mats), len(verts), len(tris)], np.uint32).tobytes())
        f.write(spheres.tobytes()); f.write(mats.tobytes())
        f.write(verts.astype(np.float32).tobytes()); f.write(tris.astype(np.uint32).tobytes())
        f.write(normals9.astype(np.float32).tobytes()); f.write(tri_mat.astype(np.uint32).tobytes())
    # (contexts > 1: RendererHIP over a group of contexts on GPU 0, tile split -- the frame must not change)
    p = subprocess.run([demo, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")] + ([str(contexts)] if contexts > 1 else []),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:] + p.stdout[-500:]
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    tail = raw[:8].view(np.uint32)
    img = raw[8:8 + w * h * 16].view(np.float32).reshape(-1, 4)
    acc = raw[8 + w * h * 16:].view(np.float32).reshape(-1, 3)
    assert tail[0] == frames + 1

    osc = orc.Scene()
    osc.set_spheres(spheres); osc.set_materials(mats); osc.set_mesh(verts, tris, normals9, tri_mat); osc.build_kd()
    assert tail[1] == len(osc.export_kd()["left"])                       # the reference builder's node count
    cam = orc.camera(70.0, 0.05, 100.0, w, h)
    o_acc = np.zeros((h * w, 3), np.float32); o_img = np.zeros((h * w, 4), np.float32)
    st = orc.Settings(1, 1 if use_spheres else 0, 1, 0, bounces)
    for fr in range(1, frames + 1):
        osc.render_frame(st, cam["pos"], cam["dirs"], w, h, fr, o_acc, o_img)
    assert np.array_equal(bits(acc), bits(o_acc))
    assert np.array_equal(bits(img), bits(o_img))


# ---- rays with exactly-zero direction components (NaN slab tests) -----------------------------------------------------------
@pytest.fixture
def parking_renderer(rt, monkeypatch):
    """A context that parks tree-unprunable bounce rays even in a tiny tree (normally only from 2^18 nodes up), so the
    whole-chip reach + test kernels and the resume launch run on the small meshes the oracle can check."""
    monkeypatch.setenv("RT_PARK_MIN_NODES", "1")
    return rt.Renderer(0)


@pytest.mark.parametrize("order", ORDERS)
@pytest.mark.parametrize("mode", [0, 2])
def test_zero_direction_rays_parked_and_resumed(rt, orc, parking_renderer, mode, order):
    """Same planted zero components as the test below, through the parking path: more such rays than parking slots, so some
    are parked and resumed and the rest are walked in the kernel, in one frame."""
    test_zero_direction_components_all_patterns(rt, orc, parking_renderer, mode, order, False)
    test_zero_direction_components_all_patterns(rt, orc, parking_renderer, mode, order, True)


@pytest.mark.parametrize("mirror_walls", [False, True], ids=["one-round", "a-round-per-bounce"])
@pytest.mark.parametrize("order", ORDERS)
def test_centre_column_through_the_glass_sphere_parks_whole_families(rt, orc, parking_renderer, order, mirror_walls):
    """Spheres and mesh together, default camera: the primary rays of the centre pixel column have direction.x == +0 and the glass
    sphere is centred on x = 0, so the refracted rays stay in the plane x = 0: dozens of bounce rays per frame that the tree
    cannot prune, all of one family (same origin.x, same zero).  They are parked, resolved together by the whole-chip kernels and
    resumed.  (From inside, the reference's sphere test only knows the near root, so such a ray leaves its sphere unnoticed and
    ends on a diffuse wall: one round.)  With perfectly smooth walls -- floor, ceiling and back wall are spheres centred on x = 0
    too -- the whole column keeps bouncing inside the plane: every resumed path parks again, a round per bounce."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(28))
    if mirror_walls:
        scene.materials["roughness"][0] = np.float32(0.0)
        osc.set_materials(scene.materials)
    got, want = run_both(rt, orc, parking_renderer, scene, osc, 96, 160, 2, 8, 3, order=order)
    assert_exact(got, want)
    col = want["ids"].reshape(160, 96)[:, 48]
    assert (col == 7).sum() > 20                           # the centre column sees the glass sphere (index 7)


@pytest.mark.parametrize("order", ORDERS)
@pytest.mark.parametrize("mode", [0, 2], ids=["mesh", "spheres+mesh"])
def test_first_hit_reuse_keeps_every_bit(rt, orc, renderer, parking_renderer, mode, order):
    """rt_settings.reuse_first_hits: the primary ray's mesh hit is found once per pixel (a launch that traces primary rays only) and
    every Render starts from it.  Ids, accumulation, image and ray count against the oracle, bit-exact: three single Renders
    (every primary ray answered from the kept hits), then 21 Renders in one call (launches of 16 and 5) against a context that
    walks every primary ray, then a moved camera (the kept hits are dropped and found again), then the same through the parking
    path (the default camera's centre column has direction.x == +0: pre-pass, parked rays and kept hits together)."""
    from raytracer_amd import scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(28), spheres=mode == 2)
    w, h = 96, 160
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, mode, 8, 3, order=order, reuse=True)
    assert_exact(got, want)
    assert got["cnt"].primary_hits_reused + got["cnt"].empty_unit_rays == 3 * w * h and got["cnt"].primary_hits_reused > 0
    # camera and scene were re-sent before every Render: the per-view data was built once (unit list in mesh mode, first hits)
    assert got["cnt"].view_setups == (2 if mode == 0 else 1)
    # deeper, several Renders per launch: against the same context walking every primary ray
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings()
    out = {}
    for reuse in (False, True):
        st.SceneMode, st.Bounces, st.TraversalOrder, st.ReuseFirstHits = mode, 8, order, reuse
        renderer.ResetFrameIndex(); renderer.reset_counters()
        renderer.Render(scene, cam, frames=21)
        out[reuse] = (renderer.read_accum().copy(), renderer.GetImage().data.copy(), renderer.read_counters())
    st.TraversalOrder, st.ReuseFirstHits = 0, False
    assert np.array_equal(bits(out[True][0]), bits(out[False][0])) and np.array_equal(bits(out[True][1]), bits(out[False][1]))
    assert out[True][2].rays == out[False][2].rays
    assert out[True][2].primary_hits_reused + out[True][2].empty_unit_rays == 21 * w * h and out[False][2].primary_hits_reused == 0
    # another viewpoint: the hits kept for the first camera must not survive
    cam2 = rt.Camera(70.0, 0.05, 100.0)
    fwd = np.array([-0.2, -0.35, -1.0], np.float32); fwd /= np.linalg.norm(fwd)
    cam2.SetPose(np.array([0.4, 1.3, 2.2], np.float32), fwd.astype(np.float32))
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, mode, 8, 2, cam=cam2, order=order, reuse=True)
    assert_exact(got, want)
    # with parking in this small tree
    got, want = run_both(rt, orc, parking_renderer, scene, osc, w, h, mode, 8, 2, order=order, reuse=True)
    assert_exact(got, want)


@pytest.mark.parametrize("order", ORDERS)
def test_empty_units_are_not_traced(rt, orc, renderer, order):
    """A scene without spheres: 16-pixel work units none of whose primary rays enters the root box are left out of the trace
    launch (their Renders add Miss()'s zero) and their TraceRay invocations are counted on the host.  A small mesh in a lot of
    sky against the oracle -- ids, accumulation, image, ray count -- in single Renders and 5 Renders per launch; the same under a
    tile split of three contexts; and a camera that looks away from the mesh (nearly every unit empty)."""
    from raytracer_amd import dist as rtd, scenes
    centre = np.array([0.3, 0.9, -1.0], np.float32)
    scene, osc = make_scene(rt, orc, mesh=scenes.cube_sphere(10, center=tuple(centre), radius=0.45), spheres=False)
    w, h = 200, 150
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, 0, 6, 3, order=order)
    assert_exact(got, want)
    assert got["cnt"].empty_unit_rays > 0.3 * 3 * w * h and (want["ids"] >= 0).sum() > 500
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = renderer.GetSettings(); st.SceneMode, st.Bounces, st.TraversalOrder = 0, 6, order
    renderer.ResetFrameIndex(); renderer.Render(scene, cam, frames=3)
    assert np.array_equal(bits(renderer.read_accum().reshape(-1, 3)), bits(want["acc"]))
    assert np.array_equal(bits(renderer.GetImage().data.reshape(-1, 4)), bits(want["img"]))
    acc = np.zeros((h, w, 3), np.float32)
    for rank in range(3):
        rr = rt.Renderer(0)
        s = rr.GetSettings(); s.SceneMode, s.Bounces, s.TraversalOrder = 0, 6, order
        rr.OnResize(w, h); rr.set_bands(rank, 3)
        for _ in range(3):
            rr.Render(scene, cam)
        rows = rtd.band_rows(h, rank, 3)
        acc[rows] = rr.read_accum()[rows]
    assert np.array_equal(bits(acc.reshape(-1, 3)), bits(want["acc"]))
    st.TraversalOrder = 0
    away = rt.Camera(70.0, 0.05, 100.0)
    away.SetPose(np.array([0.0, 1.0, 2.0], np.float32), np.array([0.0, 0.6, 0.8], np.float32))
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, 0, 6, 2, cam=away, order=order)
    assert_exact(got, want)
    # (all but the units of the centre pixel column, whose rays have direction.x == +0 and go to the pre-pass)
    assert got["cnt"].rays == 2 * w * h and got["cnt"].empty_unit_rays >= 0.9 * 2 * w * h and not want["acc"].any()


def test_first_hit_reuse_under_a_tile_split_with_the_filter(rt, orc, renderer):
    """Kept first hits with bands and the rows bordering them (3x3 filter under a tile split): three band contexts, two Renders
    each, assembled image and accumulation against the oracle's filtered frame."""
    from raytracer_amd import dist as rtd, scenes
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(24, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0))
    w, h = 150, 100
    got, want = run_both(rt, orc, renderer, scene, osc, w, h, 2, 5, 2, aa=True)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    img = np.zeros((h, w, 4), np.float32); acc = np.zeros((h, w, 3), np.float32)
    for rank in range(3):
        rr = rt.Renderer(0)
        s = rr.GetSettings(); s.SceneMode, s.Bounces, s.AntiAliasing, s.ReuseFirstHits = 2, 5, True, True
        rr.OnResize(w, h); rr.set_bands(rank, 3)
        rr.Render(scene, cam); rr.Render(scene, cam)
        rows = rtd.band_rows(h, rank, 3)
        img[rows] = rr.GetImage().data[rows]; acc[rows] = rr.read_accum()[rows]
    assert np.array_equal(bits(acc.reshape(-1, 3)), bits(want["acc"]))
    assert np.array_equal(bits(img.reshape(-1, 4)), bits(want["img"]))


@pytest.mark.parametrize("counting", [True, False], ids=["counting-kernels", "timed-kernels"])
@pytest.mark.parametrize("order", ORDERS)
@pytest.mark.parametrize("mode", [0, 2])
def test_zero_direction_components_all_patterns(rt, orc, renderer, mode, order, counting):
    """Host-supplied directions with +0 / -0 (and denormal or tiny components) planted in x, y, z (alone and combined) on scattered pixels, next to the default
    camera's own +0 column: the pre-pass (x == +0 primaries), the wave-cooperative walk (x == -0, and any bounce) and the
    ordinary walk (zeros in y / z only) must all reproduce the reference's visited sets and hits.  With the counting kernels
    (every unit traced from the root, node and triangle counters compared) and with the kernels that are timed (units that are
    not traced, primary walks that start at their unit's entry node)."""
    from raytracer_amd import scenes
    w, h = 128, 80
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(28))
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h).copy()
    rng = np.random.default_rng(5)
    pick = rng.choice(w * h, 130, replace=False)
    pz, nz = np.float32(0.0), np.float32(-0.0)
    for i, px in enumerate(pick):
        k = i % 13
        if k == 0: dirs[px, 0] = pz
        elif k == 1: dirs[px, 0] = nz
        elif k == 2: dirs[px, 1] = pz
        elif k == 3: dirs[px, 2] = nz
        elif k == 4: dirs[px, 0] = pz; dirs[px, 1] = pz
        elif k == 5: dirs[px, 0] = nz; dirs[px, 2] = pz
        elif k == 6: dirs[px, 1] = nz; dirs[px, 2] = nz
        elif k == 7: dirs[px, 0] = pz; dirs[px, 1] = nz; dirs[px, 2] = pz          # a ray going nowhere
        elif k == 8: dirs[px, 0] = pz; dirs[px] *= np.float32(3.0)                 # not unit length
        # components whose inverse is infinite or huge without being zero: the walk's box tests can meet inf - inf there as well
        # (the timed front-to-back kernels hand such rays to the reference's own walk)
        elif k == 9: dirs[px, 1] = np.float32(1e-40)                               # denormal: 1 / y = inf
        elif k == 10: dirs[px, 2] = np.float32(-1e-30)                             # 1 / z = -1e30: products overflow
        elif k == 11: dirs[px, 1] = np.float32(3e-20)                              # 1 / y just above 2^64
        else: dirs[px, 2] = np.float32(1e-19)                                      # 1 / z just below 2^64
    st = renderer.GetSettings()
    st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed = mode, 5, True, False, True, 1337
    st.TraversalOrder = order
    renderer.OnResize(w, h); renderer.ResetFrameIndex(); renderer.record_ids(True); renderer.set_stats(counting); renderer.reset_counters()
    renderer.upload(scene, cam)
    renderer.set_ray_dirs(dirs)
    try:
        acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
        cnt = orc.Counters()
        ost = orc.Settings(1, mode, 1, 0, 5)
        for f in range(1, 4):
            renderer.Render(scene, cam)
            osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img, primary_ids=ids if f == 1 else None, counters=cnt)
        got = dict(ids=renderer.read_primary_ids().ravel(), acc=renderer.read_accum().reshape(-1, 3),
                   img=renderer.GetImage().data.reshape(-1, 4), cnt=renderer.read_counters())
        # NaN never equals NaN bitwise-safe compare is what bits() does; a ray going nowhere yields NaN radiance in both
        if counting: assert_mesh(got, dict(ids=ids, acc=acc, img=img, cnt=cnt), order)
        else: assert_exact(got, dict(ids=ids, acc=acc, img=img, cnt=cnt))
    finally:
        renderer.set_ray_dirs(None); renderer.set_stats(False); st.TraversalOrder = 0
