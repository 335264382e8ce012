#!/usr/bin/env python3
"""ms per Render while the camera moves (every Render a first frame after a new rt_set_camera): bench.view_change for one or
more configurations.   python tools/view_change_probe.py 3 4 [--steps 16]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import raytracer_amd as rt
    argv = sys.argv[1:]
    steps = 16
    if "--steps" in argv:
        i = argv.index("--steps"); steps = int(argv[i + 1]); del argv[i:i + 2]
    kinds = ("rotate", "translate")
    if "--only" in argv:
        i = argv.index("--only"); kinds = (argv[i + 1],); del argv[i:i + 2]
    args = argv
    for cid in [int(a) for a in args] or [3]:
        cfg = bench.CONFIGS[cid]
        scene = bench.build_scene(cfg)
        cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(cfg["w"], cfg["h"])
        r = rt.Renderer(0); r.download = False
        st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
        r.OnResize(cfg["w"], cfg["h"]); r.upload(scene, cam)
        bench.settle_gc()                                   # (a full collection over the host copy of the scene takes ~50 ms)
        # the device idles while the host builds the scene: render for half a second first
        import time
        t_end, f = time.perf_counter() + 0.5, 1
        while time.perf_counter() < t_end:
            r.render_only(f); f += 1
            if f % 8 == 0:
                r.sync()
        r.sync()
        print(json.dumps({"config": cid, "view_change": bench.view_change(r, scene, cam, cfg, torch, steps, kinds)}), flush=True)
        del r, scene


if __name__ == "__main__":
    main()
