"""CPU-side tests of the product: the C-ABI library loads and exports what include/rt_mi355x.h declares, the host camera
and the binned KD builder reproduce the reference's host code (through the oracle and the golden fixtures), and the
library fails loudly where there is no GPU.  No compute call touches a GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, bits

GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_loads_and_exports_every_declared_symbol():
    from raytracer_amd import _capi
    L = _capi.lib()
    header = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in the header"
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_no_gpu_fails_loudly():
    """Without a device rt_create returns NULL with a message and the Python mirror raises: there is no CPU fallback."""
    import raytracer_amd as rt
    from raytracer_amd import _capi
    L = _capi.lib()
    h = L.rt_create(10_000)
    assert not h
    assert L.rt_last_error(None)
    with pytest.raises(_capi.RtError):
        rt.Renderer(10_000)


def test_product_never_imports_the_oracle():
    """The oracle is the checker, never part of the shipped path: nothing under raytracer_amd/ imports, links or loads it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(base, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(base, f)
                assert "liborc" not in src and "rt_oracle" not in src and "libref" not in src, os.path.join(base, f)


@pytest.mark.parametrize("w,h", [(64, 36), (1280, 720), (1920, 1080), (3840, 2160)])
def test_host_camera_matches_reference_camera(orc, w, h):
    import raytracer_amd as rt
    g = np.load(os.path.join(GOLD, "ref_kat.npz"))
    cam = rt.Camera(70.0, 0.05, 100.0)
    cam.OnResize(w, h)
    tag = "cam_%dx%d_" % (w, h)
    assert np.array_equal(bits(cam.GetPosition()), bits(g[tag + "pos"]))
    assert np.array_equal(bits(cam.GetDirection()), bits(g[tag + "fwd"]))
    assert np.array_equal(bits(cam.GetProjection()), bits(g[tag + "proj"]))
    assert np.array_equal(bits(cam.GetInverseProjection()), bits(g[tag + "inv_proj"]))
    assert np.array_equal(bits(cam.GetView()), bits(g[tag + "view"]))
    assert np.array_equal(bits(cam.GetInverseView()), bits(g[tag + "inv_view"]))


def test_host_camera_pose_change(orc):
    import raytracer_amd as rt
    cam = rt.Camera(55.0, 0.1, 50.0)
    cam.OnResize(200, 120)
    fwd = np.array([0.3, -0.2, -0.9], np.float32); fwd /= np.linalg.norm(fwd)
    cam.SetPose((0.4, 1.3, 2.5), fwd)
    o = orc.camera(55.0, 0.1, 50.0, 200, 120, pos=(0.4, 1.3, 2.5), fwd=fwd)
    for mine, k in ((cam.GetProjection(), "proj"), (cam.GetInverseProjection(), "inv_proj"), (cam.GetView(), "view"),
                    (cam.GetInverseView(), "inv_view")):
        assert np.array_equal(bits(mine), bits(o[k])), k


def _leaf_lists(flat):
    leaves = (flat["left"] < 0) & (flat["right"] < 0)
    return [tuple(flat["leaf_idx"][f:f + c]) for f, c in zip(flat["first"][leaves], flat["count"][leaves])]


@pytest.mark.parametrize("name", ["grid40", "cubesphere16"])
def test_binned_builder_reproduces_reference_tree(name):
    """The product's binned-counting builder yields the tree the reference's KDTreeCPU built (golden fixture)."""
    import raytracer_amd as rt
    from raytracer_amd import scenes
    g = np.load(os.path.join(GOLD, "ref_kd_%s.npz" % name))
    verts, tris, *_ = scenes.displaced_grid(40) if name == "grid40" else scenes.cube_sphere(16)
    for threads in (1, 4):
        kd = rt.KDTree(len(tris), tris, len(verts), verts, nthreads=threads)
        mine = kd.arrays()
        assert mine["depth"] == int(g["depth"][0])
        assert np.array_equal(mine["left"], g["left"]) and np.array_equal(mine["right"], g["right"])
        assert np.array_equal(bits(mine["center"]), bits(g["center"])) and np.array_equal(bits(mine["extent"]), bits(g["extent"]))
        gl = dict(left=g["left"], right=g["right"], first=g["first"], count=g["count"], leaf_idx=g["leaf_idx"])
        assert _leaf_lists(mine) == _leaf_lists(gl)


@pytest.mark.parametrize("n,drop", [(1, 0), (2, 3), (7, 0), (19, 11), (31, 1)])
def test_binned_builder_matches_oracle_builder(orc, n, drop):
    """Ragged and tiny meshes (down to a root that is itself a leaf) against the oracle's naive restatement."""
    import raytracer_amd as rt
    from raytracer_amd import scenes
    verts, tris, normals9, tri_mat, _ = scenes.displaced_grid(n, drop=drop)
    kd = rt.KDTree(len(tris), tris, len(verts), verts, nthreads=3)
    mine = kd.arrays()
    sc = orc.Scene(); sc.set_mesh(verts, tris, normals9, tri_mat); sc.build_kd()
    theirs = sc.export_kd()
    assert np.array_equal(mine["left"], theirs["left"]) and np.array_equal(mine["right"], theirs["right"])
    assert np.array_equal(bits(mine["center"]), bits(theirs["center"])) and np.array_equal(bits(mine["extent"]), bits(theirs["extent"]))
    assert _leaf_lists(mine) == _leaf_lists(theirs)
    assert mine["depth"] == theirs["depth"]


def test_builder_at_a_size_where_its_threads_share_nodes_and_subtrees(orc):
    """288,800 triangles: the root is split with its counting cut into chunks for all threads, the next levels are tasks of the
    pool, whole subtrees below are built thread by thread and numbered in parallel.  One thread, five threads and the oracle's
    naive restatement of KDTreeCPU::constructTreeStackless give the same tree, bit for bit."""
    import raytracer_amd as rt
    from raytracer_amd import scenes
    verts, tris, normals9, tri_mat, _ = scenes.displaced_grid(380, x_range=(-15.0, 15.0), z_range=(-30.0, 0.0))
    assert len(tris) >= 1 << 18
    sc = orc.Scene(); sc.set_mesh(verts, tris, normals9, tri_mat); sc.build_kd()
    theirs = sc.export_kd()
    for threads in (1, 5):
        mine = rt.KDTree(len(tris), tris, len(verts), verts, nthreads=threads).arrays()
        assert mine["depth"] == theirs["depth"]
        for k in ("left", "right"):
            assert np.array_equal(mine[k], theirs[k])
        for k in ("center", "extent"):
            assert np.array_equal(bits(mine[k]), bits(theirs[k]))
        leaves = (mine["left"] < 0) & (mine["right"] < 0)
        assert np.array_equal(mine["count"][leaves], theirs["count"][leaves])
        assert np.array_equal(np.concatenate([mine["leaf_idx"][f:f + c] for f, c in zip(mine["first"][leaves], mine["count"][leaves])]),
                              np.concatenate([theirs["leaf_idx"][f:f + c] for f, c in zip(theirs["first"][leaves], theirs["count"][leaves])]))


def test_builder_rejects_bad_input_and_honours_node_guard():
    import raytracer_amd as rt
    from raytracer_amd import _capi, scenes
    verts, tris, *_ = scenes.displaced_grid(12)
    bad = tris.copy(); bad[5, 1] = len(verts) + 7
    with pytest.raises(_capi.RtError):
        rt.KDTree(len(bad), bad, len(verts), verts)
    with pytest.raises(_capi.RtError):
        rt.KDTree(len(tris), tris, len(verts), verts, max_nodes=8)       # the tree needs far more than 8 nodes
    kd = rt.KDTree(len(tris), tris, len(verts), verts, max_nodes=100000)
    assert kd.getNumNodes() > 8


def test_builder_guard_stops_the_divergent_case():
    """A triangle fan (many boxes sharing one region) is where the reference builder grows without bound; the guard
    turns that into an error instead of an out-of-memory kill."""
    import raytracer_amd as rt
    from raytracer_amd import _capi
    k = 64
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    verts = np.concatenate([[[0, 0, 0]], np.stack([np.cos(ang), np.zeros(k), np.sin(ang)], 1)]).astype(np.float32)
    tris = np.array([[0, 1 + i, 1 + (i + 1) % k] for i in range(k)], np.uint32)
    with pytest.raises(_capi.RtError):
        rt.KDTree(len(tris), tris, len(verts), verts, max_nodes=200000, nthreads=2)


def test_host_side_under_the_sanitizers(tmp_path):
    """rt_host.cpp (KD builder on 1, 3 and 8 threads, flat view, refusal of bad indices, camera) compiled with
    -fsanitize=address,undefined and with -fsanitize=thread and driven by tests/cpp/host_sanitizers.cpp: no report, the same
    tree from every thread count."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for kind in ("address,undefined", "thread"):
        exe = str(tmp_path / ("san_" + kind.split(",")[0]))
        cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + kind, "-fno-omit-frame-pointer", "-I" + os.path.join(root, "include"),
                             os.path.join(root, "tests", "cpp", "host_sanitizers.cpp"), os.path.join(root, "raytracer_amd", "csrc", "rt_host.cpp"),
                             "-o", exe, "-pthread"], capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr[-2000:]
        run = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
        assert "WARNING" not in run.stderr and "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
        lines = [l for l in run.stdout.splitlines() if l.startswith("round ")]
        assert len(lines) == 6
        for r in range(2):                                     # the same tree from 1, 3 and 8 threads
            assert len({l.split(":", 1)[1] for l in lines[3 * r: 3 * r + 3]}) == 1
        assert "bad index: refused" in run.stdout
