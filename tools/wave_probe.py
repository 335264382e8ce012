"""Probe: per-wave start/end clocks and loop iterations of the persistent trace kernel (counting variant), to see how evenly
the waves of one frame finish.  python tools/wave_probe.py <config>"""
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
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 3       # 3 = counting kernel + wave clocks, 2 = the timed kernel + wave clocks
r.set_stats(flags); r.reset_counters()
r.render_only(4); r.sync()
c = r.read_counters()
wc = r.read_wave_clock().astype(np.float64)
start, end, it = wc[:, 0], wc[:, 1], wc[:, 2]
t0 = start.min()
span = (end.max() - t0) * 0.01
dur = (end - start) * 0.01
q = lambda a: "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))
print("config %d: %d waves, kernel span %.1f us, rays %d" % (cfgno, len(wc), span, c.rays))
print("  wave start offset us :", q((start - t0) * 0.01))
print("  wave end offset us   :", q((end - t0) * 0.01))
print("  wave duration us     :", q(dur))
print("  iterations           :", q(it))
print("  mean busy fraction (sum of wave durations / (waves x span)): %.3f" % (dur.sum() / (len(wc) * span)))
print("  lane-steps per iteration (rays / sum iterations): %.1f of 64" % (c.rays / it.sum()))
order = np.argsort(-end)
print("  last waves to finish (end us, duration us, iterations):", [(round((end[i] - t0) * 0.01, 1), round(dur[i], 1), int(it[i])) for i in order[:12]])
per_it = dur / np.maximum(it, 1)
print("  us per iteration     :", q(per_it))
late = end > t0 + 0.5 * (end.max() - t0)
print("  waves still running after half the span: %d of %d; their iterations: %s" % (late.sum(), len(wc), q(it[late])))
if wc[:, 3].sum() > 0:
    trips, lanes, walk = wc[:, 3], wc[:, 4], wc[:, 5] * 0.01
    print("  walk loop: %.0f trips over all waves, %.2f us per trip, %.3f of the waves' time" % (trips.sum(), walk.sum() / trips.sum(), walk.sum() / dur.sum()))
    print("  trips per iteration  :", q(trips / np.maximum(it, 1)))
    wait, shade = wc[:, 6] * 0.01, wc[:, 7] * 0.01
    print("  of the waves' time: %.3f waiting for paths of other waves, %.3f in the shading block, %.3f elsewhere (refill, ray set-up, hand-over)" %
          (wait.sum() / dur.sum(), shade.sum() / dur.sum(), 1.0 - (walk.sum() + wait.sum() + shade.sum()) / dur.sum()))
    print("  waiting us per wave  :", q(wait))
    if lanes.sum() > 0:
        print("  walking lanes per trip: %.1f of 64 over the frame" % (lanes.sum() / trips.sum()))
    print("  the 12 last waves (duration us, iterations, trips, us per trip, us outside the walk loop):",
          [(round(dur[i], 1), int(it[i]), int(trips[i]), round(walk[i] / max(trips[i], 1), 2), round(dur[i] - walk[i], 1)) for i in order[:12]])
    for lo, hi in [(0, 100), (100, 200), (200, 300), (300, 400), (400, 500), (500, 1e9)]:
        m = (dur >= lo) & (dur < hi)
        if m.any():
            print("    waves lasting [%g, %g) us: %d, trips per wave %.0f, us per trip %.2f" % (lo, hi, m.sum(), trips[m].mean(), walk[m].sum() / max(trips[m].sum(), 1)))
if cfg["mesh"]:
    import torch
    from raytracer_amd import dist as rtd
    steps = rtd.accum_tensor(r)[..., 3].cpu().numpy()
    print("  per-pixel path steps (nodes + triangles over all bounces):", q(steps.ravel()), " mean %.1f" % steps.mean())
    ys, xs = np.unravel_index(np.argsort(-steps.ravel())[:10], steps.shape)
    print("  heaviest pixels (x, y, steps):", [(int(x), int(y), int(steps[y, x])) for x, y in zip(xs, ys)])
    rows = steps.sum(axis=1); cols = steps.sum(axis=0)
    print("  share of all steps in the heaviest 1%% of pixels: %.3f; in rows %d..%d: %.3f" % (
        np.sort(steps.ravel())[-steps.size // 100:].sum() / steps.sum(), 0, h // 4, rows[: h // 4].sum() / steps.sum()))
    hist = np.histogram(steps.ravel(), bins=[0, 1, 2, 8, 32, 128, 512, 2048, 8192, 1 << 20])[0]
    print("  histogram of path steps [0,1,2,8,32,128,512,2048,8192+]:", hist.tolist())
