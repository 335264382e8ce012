"""Probe: what the frame hand-off costs (device -> host, 16 B per pixel): rt_read_image into pageable memory against
rt_map_image into the library's page-locked staging buffer, and the full drop-in Render() (trace + resolve + hand-off)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
import bench

for cfgno in (2, 4):
    cfg = bench.CONFIGS[cfgno]
    w, h = cfg["w"], cfg["h"]
    scene = bench.build_scene(bench.CONFIGS[2])
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    r = rt.Renderer(0)
    st = r.GetSettings(); st.SceneMode, st.Bounces = 1, 5
    r.OnResize(w, h); r.download = False
    for _ in range(3):
        r.Render(scene, cam)
    r.read_image(); r.map_image()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): r.read_image()
    t1 = time.perf_counter()
    for _ in range(n): r.map_image()
    t2 = time.perf_counter()
    r.download = True
    for _ in range(3): r.Render(scene, cam)
    t3 = time.perf_counter()
    for _ in range(n): r.Render(scene, cam)
    t4 = time.perf_counter()
    mb = w * h * 16 / 1e6
    print("%dx%d (%.0f MB/frame): rt_read_image %.2f ms (%.1f GB/s)  rt_map_image %.2f ms (%.1f GB/s)  Render() with hand-off %.2f ms/frame" % (
        w, h, mb, (t1 - t0) / n * 1e3, mb / ((t1 - t0) / n) / 1e3, (t2 - t1) / n * 1e3, mb / ((t2 - t1) / n) / 1e3, (t4 - t3) / n * 1e3))
