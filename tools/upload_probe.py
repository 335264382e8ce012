#!/usr/bin/env python3
"""Cost of re-sending the scene tables before every launch (what RaytracerLayer::Render does): tools/upload_probe.py CONFIG"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracer_amd as rt
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = bench.CONFIGS[cid]
scene = bench.build_scene(cfg)
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(cfg["w"], cfg["h"])
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
r.OnResize(cfg["w"], cfg["h"]); r.upload(scene, cam)
for per in (1, 16):
    for resend in (False, True):
        r.reset_accum(); r.render_only(1, per); r.sync()
        r.kernel_time(reset=True)
        n = 64 // per
        t0 = time.perf_counter()
        th = 0.0
        for i in range(n):
            if resend:
                t1 = time.perf_counter(); r.upload(scene, cam); th += time.perf_counter() - t1
            r.render_only(1 + i * per, per)
        t_issue = time.perf_counter() - t0
        r.sync()
        dt = time.perf_counter() - t0
        kt = r.kernel_time()
        print("config %d, %2d per launch, resend=%d: %.4f ms/frame, kernel %.4f ms/frame, host time in upload %.3f ms per launch, issue %.3f ms per launch"
              % (cid, per, resend, dt / 64 * 1e3, kt.total_ms / max(kt.frames, 1), th / n * 1e3, t_issue / n * 1e3), flush=True)
