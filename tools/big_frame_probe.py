#!/usr/bin/env python3
"""Frames beyond 4K (run on the GPU box from the repo root, ideally against the -DRT_CHECKED=1 build): 7680 x 4320 and a
ragged 8191 x 4099, the 69,938-triangle grid with and without the default spheres, one launch of 1 and one of 16 Renders;
ids / sums / image against the oracle, bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
from oracle import orc

bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
mats = scenes.default_materials()
verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(187, mat_offset=len(mats))
scene = rt.Scene(); scene.spheres = scenes.default_spheres()
scene.set_triangles(verts, tris, normals9, tri_mat, nthreads=16)
scene.materials = np.concatenate([mats, mesh_mats])
osc = orc.Scene(); osc.set_spheres(scene.spheres); osc.set_materials(scene.materials)
osc.set_mesh(verts, tris, normals9, tri_mat); osc.import_kd(scene.kd_tree.arrays())
r = rt.Renderer(0)
bad = 0
for (w, h), mode, per_launch in (((7680, 4320), 0, 1), ((8191, 4099), 2, 1), ((7680, 4320), 1, 1), ((8191, 4099), 0, 16)):
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    st = r.GetSettings(); st.SceneMode, st.Bounces = mode, 3
    r.OnResize(w, h); r.ResetFrameIndex(); r.record_ids(True); r.reset_counters()
    t0 = time.time(); r.Render(scene, cam, frames=per_launch); g_cnt = r.read_counters(); t1 = time.time()
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    acc = np.zeros((h * w, 3), np.float32); img = np.zeros((h * w, 4), np.float32); ids = np.zeros(h * w, np.int32)
    ost = orc.Settings(1, mode, 1, 0, 3)
    for f in range(1, per_launch + 1):
        osc.render_frame(ost, cam.GetPosition(), dirs, w, h, f, acc, img, seed=st.Seed, primary_ids=ids if f == 1 else None, nthreads=16)
        print("  oracle frame %d" % f, flush=True)
    t2 = time.time()
    ok = (np.array_equal(r.read_primary_ids().ravel(), ids) and np.array_equal(bits(r.read_accum().reshape(-1, 3)), bits(acc))
          and np.array_equal(bits(r.GetImage().data.reshape(-1, 4)), bits(img)))
    bad += not ok
    print("%s %d x %d, mode %d, %d Renders in one launch: library %.2f s, oracle %.1f s, rays %d" % ("ok      " if ok else "MISMATCH", w, h, mode, per_launch, t1 - t0, t2 - t1, g_cnt.rays), flush=True)
sys.exit(1 if bad else 0)
