#!/usr/bin/env python3
"""Scene set-up times on the GPU box (run from the repo root): mesh generation, KD build (rt_host_kd_build) and the upload
(rt_set_mesh + rt_set_kdtree: the eight node arrays, leaf records, near-tie bounds) of bench.py's configurations 3 and 4."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, raytracer_amd as rt
from raytracer_amd import scenes
for cfgi in (3, 4):
    cfg = bench.CONFIGS[cfgi]
    k = cfg["mesh"][2] if len(cfg["mesh"]) > 2 else 1.0
    t0 = time.time()
    verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(cfg["mesh"][1], x_range=(-1.5 * k, 1.5 * k), z_range=(-3.0 * k, 0.0), mat_offset=len(scenes.default_materials()))
    t1 = time.time()
    scene = rt.Scene(); scene.spheres = scenes.default_spheres()
    scene.set_triangles(verts, tris, normals9, tri_mat, nthreads=min(16, bench.host_threads()))
    scene.materials = np.concatenate([scenes.default_materials(), mesh_mats])
    t2 = time.time()
    r = rt.Renderer(0); cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(cfg["w"], cfg["h"])
    st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
    r.OnResize(cfg["w"], cfg["h"])
    t3 = time.time(); r.upload(scene, cam); t4 = time.time()
    r.render_only(1); r.sync(); t5 = time.time()
    r.render_only(2); r.sync(); t6 = time.time()
    print("config %d: %d triangles, mesh %.2f s, KD build %.2f s, upload %.2f s, first Render %.3f s, second %.4f s" % (
        cfgi, len(tris), t1 - t0, t2 - t1, t4 - t3, t5 - t4, t6 - t5), flush=True)
