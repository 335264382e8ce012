"""Probe: when the workgroups of the sphere-scene trace kernel finish (per-workgroup first / last wave end), to tell a drain
inside every workgroup from an imbalance between workgroups.  python tools/wg_probe.py [config]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
import bench

cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = bench.CONFIGS[cfgno]
scene = bench.build_scene(cfg)
w, h = cfg["w"], cfg["h"]
cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
r = rt.Renderer(0); r.download = False
st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
r.OnResize(w, h); r.upload(scene, cam)
for f in range(1, 4):
    r.render_only(f)
r.set_stats(2); r.reset_counters()
r.render_only(4); r.sync()
wc = r.read_wave_clock().astype(np.float64)
per = 16 if cfg["mode"] == 1 else int(os.environ.get("RT_PROBE_WAVES_PER_WG", "16"))
n = len(wc) // per * per
start, end, it = wc[:n, 0].reshape(-1, per), wc[:n, 1].reshape(-1, per), wc[:n, 2].reshape(-1, per)
t0 = start.min()
first, last = (end.min(axis=1) - t0) * 0.01, (end.max(axis=1) - t0) * 0.01
q = lambda a: "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))
print("config %d: %d workgroups of %d waves" % (cfgno, len(first), per))
print("  first wave of a workgroup ends (us):", q(first))
print("  last wave of a workgroup ends (us) :", q(last))
print("  last - first within a workgroup    :", q(last - first))
print("  iterations summed over a workgroup :", q(it.sum(axis=1)))
half = len(first) // 2
print("  last wave ends, first half of the grid:", q(last[:half]), "| second half:", q(last[half:]))
for x in range(8):
    print("  workgroups with blockIdx %% 8 == %d end: %s" % (x, q(last[x::8])))
