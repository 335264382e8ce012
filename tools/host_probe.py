"""Probe: host time per Render call against GPU time (python tools/host_probe.py <config> [scene mode])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt, bench
cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(bench.CONFIGS[cfgno])
if len(sys.argv) > 2: cfg["mode"] = int(sys.argv[2])
scene = bench.build_scene(cfg); w, h = cfg["w"], cfg["h"]
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
r.OnResize(w, h); r.upload(scene, cam)
for f in range(1, 4): r.render_only(f)
r.sync()
calls = []
t0 = time.perf_counter()
for f in range(4, 14):
    a = time.perf_counter(); r.render_only(f); calls.append((time.perf_counter() - a) * 1e3)
r.sync()
print("config %d: %.2f ms per frame over 10 frames; time inside each render call (ms): %s" % (cfgno, (time.perf_counter() - t0) * 100, ["%.2f" % c for c in calls]))
