#!/usr/bin/env python3
"""A tree configuration of bench.py (default 5: spheres + 10 M triangles at 4K): ms per frame with one Render per launch and with
16 per launch, printed as it goes (run on the GPU box from the repo root; PROBE_BOUNCES=<n> overrides the bounce count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracer_amd as rt

cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 5]
t0 = time.perf_counter(); scene = bench.build_scene(cfg); print("scene %.1f s" % (time.perf_counter() - t0), flush=True)
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(cfg["w"], cfg["h"])
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], int(os.environ.get("PROBE_BOUNCES", cfg["bounces"]))
r.OnResize(cfg["w"], cfg["h"])
t0 = time.perf_counter(); r.upload(scene, cam); r.sync(); print("upload %.1f s" % (time.perf_counter() - t0), flush=True)
for per in (1, 1, 16, 16):
    r.reset_accum(); r.reset_counters(); r.kernel_time(reset=True)
    t0 = time.perf_counter()
    n = 4 if per == 1 else 1
    for i in range(n):
        r.render_only(1 + i * per, per)
    r.sync()
    dt = time.perf_counter() - t0
    kt = r.kernel_time(); c = r.read_counters()
    print("%2d per launch: %.3f ms/frame, trace kernel %.3f ms/frame, pre-pass %.3f ms/frame, %.0f rays/frame, %.1f near-tie repeats/frame" %
          (per, dt / (n * per) * 1e3, kt.total_ms / max(kt.frames, 1), kt.prepass_total_ms / max(kt.frames, 1), c.rays / (n * per),
           c.near_ties / (n * per)), flush=True)
