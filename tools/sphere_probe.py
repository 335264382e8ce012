"""Probe: sphere-scene trace kernel time per frame for several frames-per-launch values (how much of a frame is ramp and tail)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
import bench

cfg = bench.CONFIGS[2]
scene = bench.build_scene(cfg)
w, h = cfg["w"], cfg["h"]
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
r.OnResize(w, h); r.upload(scene, cam)
for fpl in (1, 2, 4, 16):
    for f in range(3):
        r.render_only(1 + f * fpl, fpl)
    r.sync(); r.kernel_time(reset=True)
    n = 32 // fpl if fpl < 16 else 4
    for f in range(n):
        r.render_only(1 + f * fpl, fpl)
    r.sync(); kt = r.kernel_time()
    print("blocks/CU %s frames/launch %2d: %.1f us per frame (kernel %.1f us x %d launches)" % (
        os.environ.get("RT_TRACE_BLOCKS_PER_CU", "auto"), fpl, kt.total_ms / kt.launches / fpl * 1e3, kt.total_ms / kt.launches * 1e3, kt.launches))
