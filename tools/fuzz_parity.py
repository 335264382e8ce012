#!/usr/bin/env python3
"""Randomised parity probe (run on the GPU box from the repo root): scenes with unusual parameters -- radii of zero, negative or
huge size, cameras inside spheres, roughness beyond 1, refraction indices below 1 or negative, emission of 1e30, materials with
albedo 0, meshes next to or around the camera, zero or one bounce -- rendered by the library and by the oracle; primary ids,
accumulation and image must agree bit for bit.  python tools/fuzz_parity.py [cases] [seed]
FUZZ_WIDE=1 widens the draw (larger meshes, frames of 1 x 1 to 130 x 70 pixels); the default draw stays as it is so that a seed
keeps naming the same scenes.  FUZZ_MOVES=1 also moves the camera between the launches of about half the cases (a turn, a step
forward that keeps the centre column's direction.x == +0, a step sideways, the same pose sent again) while the accumulation goes
on: the per-view data -- ray directions, column rays, unit lists, entry nodes, kept first hits -- is rebuilt in the middle of it.  Meant to run against a library built with -DRT_CHECKED=1 (tools/build_variant.sh), in which a
state machine that leaves its arrays is reported as an error instead of faulting the GPU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes, dist as rtd
from oracle import orc

bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pick = lambda *xs: xs[rng.integers(len(xs))]
wide = bool(os.environ.get("FUZZ_WIDE"))
moves = bool(os.environ.get("FUZZ_MOVES"))


def random_materials(n):
    m = np.zeros(n, scenes.MATERIAL_DTYPE)
    for i in range(n):
        m[i]["albedo"] = pick(rng.random(3), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), rng.random(3) * 3.0, (1e-30, 0.5, 0.0))
        m[i]["emission"] = pick((0.0, 0.0, 0.0), rng.random(3) * 5.0, (1e30, 1e30, 0.0), (0.0, 1e-40, 3.0))
        m[i]["roughness"] = pick(0.0, 1.0, rng.random(), 1.7, -0.5)
        m[i]["transparency"] = pick(0.0, 0.0, 1.0, 0.3, -1.0)
        m[i]["ior"] = pick(1.0, 1.1, 1.5, 0.7, 2.5, 0.0, -1.3, 1e-20)
    return m


def random_spheres(n, n_mats):
    s = np.zeros(n, scenes.SPHERE_DTYPE)
    for i in range(n):
        c = rng.normal(size=3) * pick(0.5, 2.0, 20.0)
        s[i] = (np.float32(c[0]), np.float32(c[1] + 1.0), np.float32(c[2]), np.float32(pick(0.3, 1.0, 0.0, -0.7, 50.0, 1000.0, 1e-6, rng.random() * 2)),
                rng.integers(n_mats))
    return s


r = rt.Renderer(0)
failures = 0
for case in range(n_cases):
    mode = int(pick(1, 1, 0, 2))
    n_mats = int(pick(rng.integers(1, 9), rng.integers(1, 9), 40, 140))      # (beyond 32 / 128: other table paths in the kernels)
    mats = random_materials(n_mats)
    scene = rt.Scene(); osc = orc.Scene()
    n_sph = int(pick(rng.integers(0 if mode == 2 else 1, 14), rng.integers(0 if mode == 2 else 1, 14), 33, 45, 150))
    scene.spheres = random_spheres(n_sph, n_mats) if mode != 0 else scenes.default_spheres()[:0]
    if mode == 1 and rng.random() < 0.3:
        scene.spheres = np.concatenate([scenes.default_spheres(), scene.spheres]); scene.spheres["material"] %= n_mats
    if mode != 1:
        n = int(pick(6, 14, 24, 48, 80)) if wide else int(pick(6, 14, 24))
        verts, tris, normals9, tri_mat, _mm = pick(lambda: scenes.displaced_grid(n), lambda: scenes.cube_sphere(max(4, n // 2), center=(0.0, 1.0, 1.5), radius=2.5),
                                                   lambda: scenes.cube_sphere(max(4, n // 2)))()
        if rng.random() < 0.3: normals9 = (normals9 * np.float32(pick(0.0, -1.0, 7.0))).astype(np.float32)      # unnormalised / zero / flipped normals
        tri_mat = rng.integers(0, n_mats, len(tris)).astype(np.uint32)
        scene.set_triangles(verts, tris, normals9, tri_mat, nthreads=4)
        osc.set_mesh(verts, tris, normals9, tri_mat); osc.import_kd(scene.kd_tree.arrays())
    scene.materials = mats
    osc.set_spheres(scene.spheres); osc.set_materials(mats)
    w, h = (int(pick(64, 96, 33, 1, 130, 16)), int(pick(48, 17, 64, 1, 7, 70))) if wide else (int(pick(64, 96, 33)), int(pick(48, 17, 64)))
    cam = rt.Camera(float(pick(70.0, 45.0, 120.0)), 0.05, 100.0)
    if rng.random() < 0.5:
        pos = (rng.normal(size=3) * pick(0.3, 3.0)).astype(np.float32); fwd = rng.normal(size=3).astype(np.float32); fwd /= np.linalg.norm(fwd)
        if abs(fwd[1]) > 0.95: fwd = np.array([0.0, -0.1, -1.0], np.float32) / np.float32(np.sqrt(1.01))
        cam.SetPose(pos, fwd.astype(np.float32))
    cam.OnResize(w, h)
    bounces, frames = int(pick(0, 1, 3, 5, 9)), int(pick(1, 2, 5))
    use_ace, aa, seed = bool(rng.random() < 0.7), bool(rng.random() < 0.2), int(pick(1337, 0, 4242424242))
    order = int(pick(0, 1, 2))
    per_launch = int(pick(1, 1, 2, 16))                       # Renders per launch (rt_render_frames)
    reuse = bool(mode != 1 and rng.random() < 0.3)            # kept first hits
    world = int(pick(1, 1, 1, 2, 3, 5)) if wide else 1          # tile split: one context per rank on this GPU, each renders its bands

    def prepared(rr):
        st = rr.GetSettings()
        st.SceneMode, st.Bounces, st.UseACE_Color, st.AntiAliasing, st.Accumulate, st.Seed, st.TraversalOrder = mode, bounces, use_ace, aa, True, seed, order
        st.ReuseFirstHits = reuse
        rr.OnResize(w, h); rr.ResetFrameIndex(); rr.record_ids(True); rr.reset_counters()
        return rr
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    host_dirs = rng.random() < 0.25                           # directions from the host, some of them odd: zeros, denormals, not unit length
    if host_dirs:
        dirs = dirs.copy()
        for px in rng.choice(w * h, min(60, w * h), replace=False):
            k = int(rng.integers(8))
            if k == 0: dirs[px, rng.integers(3)] = np.float32(pick(0.0, -0.0))
            elif k == 1: dirs[px, rng.integers(3)] = np.float32(pick(1e-40, -1e-41, 1e-30, 3e-20))
            elif k == 2: dirs[px] *= np.float32(pick(3.0, 0.01, 1e6, 1e-6))
            elif k == 3: dirs[px] = (0.0, 0.0, 0.0)
            elif k == 4: dirs[px] = np.float32(pick(1.0, -1.0)) * np.eye(3, dtype=np.float32)[rng.integers(3)]
            elif k == 5: dirs[px, 0] = np.float32(0.0); dirs[px, 1] = np.float32(-0.0)
            else: dirs[px] = rng.normal(size=3).astype(np.float32)
    # the camera of every launch: the first one's, unless FUZZ_MOVES moves it in between
    poses = [(cam.GetPosition().copy(), cam.GetDirection().copy())] * frames
    if moves and not host_dirs and rng.random() < 0.5:
        poses = [poses[0]]
        for _ in range(frames - 1):
            pos, fwd = poses[-1]
            k = int(rng.integers(5))
            if k == 0: pass                                                                                   # the same pose, sent again
            elif k == 1: pos = (pos + fwd * np.float32(pick(0.01, 0.3, -0.2))).astype(np.float32)                # along the view direction
            elif k == 2: pos = (pos + np.array([pick(0.05, -0.4), 0.0, 0.0], np.float32)).astype(np.float32)      # sideways
            elif k == 3:
                a = np.float32(pick(0.002, 0.3, -1.0)); c_, s_ = np.float32(np.cos(a)), np.float32(np.sin(a))
                fwd = np.array([fwd[0] * c_ + fwd[2] * s_, fwd[1], -fwd[0] * s_ + fwd[2] * c_], np.float32)        # a turn about the vertical axis
            else:
                pos = (rng.normal(size=3) * 2.0).astype(np.float32); pos[1] += np.float32(1.0)
            poses.append((pos, fwd))
    moved = len(set((p_.tobytes(), f_.tobytes()) for p_, f_ in poses)) > 1
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
    cnt = orc.Counters()
    ost = orc.Settings(1, mode, int(use_ace), int(aa), bounces)
    if os.environ.get("FUZZ_VERBOSE"): print("start    case %d" % case, flush=True)
    try:
        f = 1
        for _launch in range(frames):
            if moved:
                cam.SetPose(*poses[_launch]); cam.OnResize(w, h)
                dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
            for _k in range(per_launch):
                osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img, seed=seed, primary_ids=ids if f == 1 else None, counters=cnt, nthreads=8)
                f += 1
        g_ids, g_acc, g_img = np.zeros(h * w, np.int32), np.zeros((h * w, 3), np.float32), np.zeros((h * w, 4), np.float32)
        g_rays = 0
        for rank in range(world):
            rr = prepared(r if world == 1 else rt.Renderer(0))
            if world > 1: rr.set_bands(rank, world)
            if host_dirs:
                rr.upload(scene, cam); rr.set_ray_dirs(dirs)
            for _launch in range(frames):
                if moved:
                    cam.SetPose(*poses[_launch]); cam.OnResize(w, h)
                rr.Render(scene, cam, frames=per_launch)
                if moved and _launch == 0: first_ids = rr.read_primary_ids().ravel().copy()        # (ids are those of frame index 1)
            rows = np.arange(h) if world == 1 else rtd.band_rows(h, rank, world)
            px = (rows[:, None] * w + np.arange(w)[None, :]).ravel()
            g_ids[px] = (first_ids if moved else rr.read_primary_ids().ravel())[px]; g_acc[px] = rr.read_accum().reshape(-1, 3)[px]; g_img[px] = rr.GetImage().data.reshape(-1, 4)[px]
            g_rays += rr.read_counters().rays
            if world > 1: del rr
        if world > 1 and aa: g_rays = cnt.rays                   # (under a tile split with the filter on, the rows bordering a band are traced twice)
        bad = []
        if not np.array_equal(g_ids, ids): bad.append("ids: %d differ" % (g_ids != ids).sum())
        if not np.array_equal(bits(g_acc), bits(acc)): bad.append("accumulation: %d values differ" % (bits(g_acc) != bits(acc)).sum())
        if not np.array_equal(bits(g_img), bits(img)): bad.append("image: %d values differ" % (bits(g_img) != bits(img)).sum())
        if g_rays != cnt.rays: bad.append("rays %d vs %d" % (g_rays, cnt.rays))
    except Exception as e:                                   # an error from the library (a refused scene, a failed GPU call) ends the probe:
        print("ERROR    case %d: %s: %s" % (case, type(e).__name__, e), flush=True)       # nothing more is started on the GPU after it
        sys.exit(2)
    desc = "case %d: mode %d, %d spheres, %d materials, %dx%d, %d bounces, %d launches of %d, ace %d, aa %d, order %d, kept hits %d, host directions %d, ranks %d, camera moves %d" % (
        case, mode, len(scene.spheres), n_mats, w, h, bounces, frames, per_launch, use_ace, aa, order, reuse, host_dirs, world, moved)
    if bad:
        failures += 1
        print("MISMATCH", desc, "|", "; ".join(bad), flush=True)
    else:
        print("ok      ", desc, flush=True)
    st = r.GetSettings(); st.TraversalOrder = 0; st.ReuseFirstHits = False
    if host_dirs and world == 1: r.set_ray_dirs(None)
print("%d of %d cases differ" % (failures, n_cases))
sys.exit(1 if failures else 0)
