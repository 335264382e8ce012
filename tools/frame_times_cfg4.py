import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import raytracer_amd as rt, bench
cfg = bench.CONFIGS[4]; scene = bench.build_scene(cfg); w, h = cfg["w"], cfg["h"]
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
r.OnResize(w, h); r.upload(scene, cam)
for f in range(1, 81):
    r.kernel_time(reset=True)
    r.render_only(f); r.sync()
    kt = r.kernel_time()
    if f <= 4 or f % 8 == 0: print("frame %d trace %.2f ms prepass %.2f ms" % (f, kt.total_ms, kt.prepass_total_ms), flush=True)
