"""Several GPUs from one process (rt_group_*, the C ABI a C++ caller uses), exercised on one GPU by listing device 0 several
times: the contexts, their streams, the pushes into context 0's memory and the event ordering are the same; only the peer
access between different devices is not exercised here.

Tile split: the assembled frame and accumulation equal the oracle's bit for bit (every pixel has one owner).  Sample split:
the contexts' sums are added in context order; against the oracle's frame-by-frame sum that is a re-association, bounded by
|d| <= 1e-4 + 1e-4 |ref|, and equal bit for bit to adding the contexts' own buffers in that order on the host."""
import numpy as np
import pytest

from conftest import bits
from test_gpu_parity import make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    return raytracer_amd


def oracle_frames(orc, osc, cam, w, h, mode, bounces, frames, aa=False):
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
    st = orc.Settings(1, mode, 1, int(aa), bounces)
    for f in range(1, frames + 1):
        osc.render_frame(st, cam.GetPosition(), dirs, w, h, f, acc, img)
    return acc, img


@pytest.mark.parametrize("mode,aa", [(1, False), (2, False), (2, True)], ids=["spheres", "spheres+mesh", "spheres+mesh+filter"])
def test_group_tile_split_equals_the_one_gpu_frame(rt, orc, mode, aa):
    from raytracer_amd import scenes
    mesh = None if mode == 1 else scenes.displaced_grid(24, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    w, h, frames = 150, 100, 3
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    want_acc, want_img = oracle_frames(orc, osc, cam, w, h, mode, 5, frames, aa)
    g = rt.GroupRenderer([0, 0, 0], split="tiles")
    st = g.GetSettings(); st.SceneMode, st.Bounces, st.AntiAliasing = mode, 5, aa
    g.OnResize(w, h)
    for _ in range(frames):
        g.Render(scene, cam)
    assert g.GetFrameIndex() == frames + 1
    assert np.array_equal(bits(g.read_accum().reshape(-1, 3)), bits(want_acc))
    assert np.array_equal(bits(g.GetImage().data.reshape(-1, 4)), bits(want_img))


@pytest.mark.parametrize("mode", [1, 0], ids=["spheres", "mesh"])
def test_group_sample_split_sums_in_context_order(rt, orc, mode):
    from raytracer_amd import scenes
    mesh = None if mode == 1 else scenes.displaced_grid(24, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0)
    scene, osc = make_scene(rt, orc, mesh=mesh)
    w, h, n, per = 128, 80, 3, 2
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    g = rt.GroupRenderer([0] * n, split="samples", frames_per_device=per)
    st = g.GetSettings(); st.SceneMode, st.Bounces = mode, 5
    g.OnResize(w, h)
    for call in (1, 2):                                           # the second call continues the accumulation
        g.Render(scene, cam)
        total = call * n * per
        assert g.GetFrameIndex() == total + 1
        want_acc, want_img = oracle_frames(orc, osc, cam, w, h, mode, 5, total)
        got = g.read_accum().reshape(-1, 3)
        assert np.all(np.abs(got - want_acc) <= 1e-4 + 1e-4 * np.abs(want_acc))
        own = [m.read_accum().reshape(-1, 3) for m in g.members]
        by_hand = own[0].copy()
        for part in own[1:]:
            by_hand = by_hand + part                              # ((a0 + a1) + a2): the order the library promises
        assert np.array_equal(bits(got), bits(by_hand))
        img = g.GetImage().data.reshape(-1, 4)
        assert np.all(np.abs(img - want_img) <= 2e-3)            # the displayed frame, from the re-associated sum
    # each context accumulated exactly its own block of frame indices
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    ost = orc.Settings(1, mode, 1, 0, 5)
    for i, m in enumerate(g.members):
        acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32)
        for call in (0, 1):
            for k in range(per):
                f = 1 + call * n * per + i * per + k
                osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img)
        # the oracle starts from zero only at frame index 1; the other blocks start from a zeroed buffer as well
        assert np.array_equal(bits(m.read_accum().reshape(-1, 3)), bits(acc)), "context %d" % i


@pytest.mark.parametrize("split", ["tiles", "samples"])
def test_group_issues_on_every_context_before_it_waits_for_any(rt, orc, split, monkeypatch):
    """In a tree large enough for parking (here forced: RT_PARK_MIN_NODES=1) every trace launch is followed by a wait for one
    word, the number of paths it parked.  A group issues the launch on all its contexts first and waits afterwards, so its GPUs
    trace side by side: when the issue phase of a call returns, the first context's kernel is still running
    (rt_group_issue_overlap; three contexts on one GPU, a frame of several hundred microseconds).  Bits as ever."""
    from raytracer_amd import scenes
    monkeypatch.setenv("RT_PARK_MIN_NODES", "1")
    scene, osc = make_scene(rt, orc, mesh=scenes.displaced_grid(60, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), mat_offset=0))
    w, h, per = 640, 360, 2
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    g = rt.GroupRenderer([0, 0, 0], split=split, frames_per_device=per)
    L = g._L
    assert L.rt_group_issue_overlap(g._g) == -1
    st = g.GetSettings(); st.SceneMode, st.Bounces = 2, 6
    g.OnResize(w, h)
    g.Render(scene, cam)
    overlaps = [L.rt_group_issue_overlap(g._g)]
    g.Render(scene, cam)
    overlaps.append(L.rt_group_issue_overlap(g._g))
    # contexts 1 and 2 were issued while context 0's launch was still on the device (every one of the three parks: mode 2)
    assert max(overlaps) >= 1, overlaps
    frames = 2 if split == "tiles" else 2 * 3 * per
    want_acc, want_img = oracle_frames(orc, osc, cam, w, h, 2, 6, frames)
    got = g.read_accum().reshape(-1, 3)
    if split == "tiles":
        assert np.array_equal(bits(got), bits(want_acc))
        assert np.array_equal(bits(g.GetImage().data.reshape(-1, 4)), bits(want_img))
    else:
        assert np.all(np.abs(got - want_acc) <= 1e-4 + 1e-4 * np.abs(want_acc))


def test_group_errors(rt):
    from raytracer_amd import _capi
    with pytest.raises(_capi.RtError):
        rt.GroupRenderer([0, 99], split="tiles")


def test_rt_bench_headless_cpp_host():
    """raytracer_amd/lib/rt_bench (plain C++ over the C ABI): a small mesh scene through its option set -- single Renders, several
    Renders per launch, kept first hits, the frame copied to the host, two contexts in a tile split.  The same frames are
    rendered every time, so the device's ray count per frame must be the same in all of them."""
    import json, os, subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "raytracer_amd", "lib", "rt_bench")
    assert os.path.exists(exe), "rt_bench is built by raytracer_amd/csrc/Makefile"
    base = [exe, "--config", "3", "--width", "320", "--height", "200", "--grid", "150", "--frames", "8", "--warmup", "1"]
    runs = {"single": [], "x4": ["--frames-per-launch", "4"], "reuse": ["--frames-per-launch", "4", "--reuse-first-hits", "1"],
            "d2h": ["--d2h", "2"], "moving": ["--moving", "1"], "tiles": ["--devices", "0,0", "--split", "tiles", "--frames-per-launch", "2"]}
    out = {}
    for name, extra in runs.items():
        p = subprocess.run(base + extra, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, (name, p.stderr[-400:])
        out[name] = json.loads(p.stdout.strip().splitlines()[-1])
        assert out[name]["frames"] == 8 and out[name]["ms_per_frame"] > 0 and out[name]["mrays_per_s"] > 0
        assert out[name]["trace_kernel_ms_per_frame_gpu0"] > 0, name          # (some launch of every run is a timed one)
    assert out["moving"]["moving"] == 1 and out["moving"]["rays_per_frame"] >= 320 * 200
    rays = {name: o["rays_per_frame"] for name, o in out.items() if name != "moving"}           # (a turning camera sees other rays)
    assert len(set(rays.values())) == 1 and rays["single"] >= 320 * 200, rays
    assert out["reuse"]["primary_hits_reused_per_frame"] > 0 and out["single"]["primary_hits_reused_per_frame"] == 0
    assert out["reuse"]["primary_hits_reused_per_frame"] + out["reuse"]["empty_unit_rays_per_frame"] == 320 * 200
    assert out["tiles"]["devices"] == 2 and out["tiles"]["split"] == "tiles"
