#!/usr/bin/env python3
"""What ray coherence buys the tree walk: tools/coherence_probe.py CONFIG [LAUNCHES] (run on the GPU box from the repo root).
Times launches of 16 Renders with 1 bounce (primary rays only: the 64 lanes of a wave hold four 8x2 pixel strips of one tile,
one origin, neighbouring directions, one sign octant) and with the configuration's own bounce count (the same primaries plus
scattered rays of random origin and octant), counts the node and triangle steps of both with the counting kernels, and prints
the trace kernel's time per step for the primaries and for the scattered rays (difference of the two runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracer_amd as rt

cid = int(sys.argv[1]); launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
PER = 16
cfg = bench.CONFIGS[cid]
scene = bench.build_scene(cfg)
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(cfg["w"], cfg["h"])
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode = cfg["mode"]
r.OnResize(cfg["w"], cfg["h"])
res = {}
for bounces in (1, cfg["bounces"]):
    st.Bounces = bounces
    r.upload(scene, cam); r.push_settings()
    r.reset_accum(); r.render_only(1001, PER); r.sync()
    r.reset_accum(); r.reset_counters(); r.kernel_time(reset=True)
    for i in range(launches):
        r.render_only(1 + i * PER, PER)
    r.sync()
    kt, cnt = r.kernel_time(), r.read_counters()
    ms = kt.total_ms / max(kt.frames, 1)
    r.set_stats(True); r.reset_accum(); r.reset_counters()
    for f in (1, 2):
        r.render_only(f)
    c2 = r.read_counters(); r.set_stats(False)
    steps = ((c2.nodes_visited - c2.prepass_nodes) + (c2.tris_tested - c2.prepass_tris)) / 2.0
    res[bounces] = (ms, steps, cnt.rays / (launches * PER))
    print("config %d, %d bounce(s): trace kernel %.4f ms/frame, %.3f M rays/frame, %.2f M node+triangle steps/frame, %.2f ns/step"
          % (cid, bounces, ms, res[bounces][2] / 1e6, steps / 1e6, ms * 1e6 / steps), flush=True)
(m1, s1, r1), (m8, s8, r8) = res[1], res[cfg["bounces"]]
print("config %d: primaries %.2f ns/step (%.2f M steps); scattered rays %.2f ns/step (%.2f M steps, %.3f M rays)"
      % (cid, m1 * 1e6 / s1, s1 / 1e6, (m8 - m1) * 1e6 / max(s8 - s1, 1), (s8 - s1) / 1e6, (r8 - r1) / 1e6), flush=True)
