#!/usr/bin/env python3
"""Only launches of one kind, for profiler passes: tools/tp_run.py CONFIG FRAMES_PER_LAUNCH LAUNCHES (run on the GPU box from the
repo root).  Every trace-kernel launch of the process is a timed one, so per-kernel counter means are not mixed with warm-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracer_amd as rt

cid, per, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = bench.CONFIGS[cid]
scene = bench.build_scene(cfg)
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(cfg["w"], cfg["h"])
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], int(os.environ.get("PROBE_BOUNCES", cfg["bounces"]))
r.OnResize(cfg["w"], cfg["h"]); r.upload(scene, cam); r.sync()
t0 = time.perf_counter()
for i in range(n):
    r.render_only(1 + i * per, per)
r.sync()
dt = time.perf_counter() - t0
kt = r.kernel_time()
print("config %d, %d x %d frames: %.3f ms/frame, trace kernel %.3f ms/frame" % (cid, n, per, dt / (n * per) * 1e3, kt.total_ms / max(kt.frames, 1)), flush=True)
