#!/usr/bin/env python3
"""Where a Render with the frame hand-off spends its time: host-side duration of rt_render and of rt_map_image /
rt_map_image_pipelined per step, for the default sphere scene at 1080p (run on the GPU box from the repo root)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes

w, h = 1920, 1080
scene = rt.Scene(); scene.spheres = scenes.default_spheres(); scene.materials = scenes.default_materials()
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = 1, 5
r.OnResize(w, h); r.upload(scene, cam)
for pipelined in (False, True):
    for f in range(1, 6):
        r.render_only(f); r.map_image(pipelined)
    r.sync()
    tr, tm = [], []
    t_all = time.perf_counter()
    for f in range(1, 41):
        t0 = time.perf_counter(); r.render_only(f); t1 = time.perf_counter(); r.map_image(pipelined); t2 = time.perf_counter()
        tr.append(t1 - t0); tm.append(t2 - t1)
    r.sync()
    t_all = time.perf_counter() - t_all
    print("pipelined=%d: %.3f ms/step; rt_render call %.3f ms, map call %.3f ms (medians)" %
          (pipelined, t_all / 40 * 1e3, np.median(tr) * 1e3, np.median(tm) * 1e3))
