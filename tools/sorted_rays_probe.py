#!/usr/bin/env python3
"""What could an ordering of the scattered rays buy?  An upper-bound experiment.

One frame's rays (every ray that walks the tree, written down by the counting kernel in the order the waves started them) are
walked again as a list by the timed walk (rt_debug_walk_rays): (a) in that order, (b) sorted on the host by direction octant
and Morton code of the origin's cell (1024^3 cells over the rays' bounding box), (c) shuffled.  Kernel times by HIP events; the
hits of the three orders are compared ray by ray.  Run under `rocprofv3 --pmc ...` with --pmc the two orders appear as
rt_probe_walk_kernel<0> (path order) and <1> (sorted).

    python tools/sorted_rays_probe.py 3 4 [--repeats 5] [--pmc]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def morton3(ix, iy, iz):
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0x3FF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x30000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x9249249)
        return v
    return spread(ix) | (spread(iy) << np.uint64(1)) | (spread(iz) << np.uint64(2))


def sort_key(rays):
    o, d = rays[:, 0:3], rays[:, 4:7]
    octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2)))
    lo, hi = o.min(axis=0), o.max(axis=0)
    cell = np.clip(((o - lo) / np.maximum(hi - lo, 1e-20) * 1023.0), 0, 1023).astype(np.uint32)
    return (octant << np.uint64(30)) | morton3(cell[:, 0], cell[:, 1], cell[:, 2])


def main():
    import raytracer_amd as rt
    from raytracer_amd import _capi
    argv = sys.argv[1:]
    pmc = "--pmc" in argv
    if pmc: argv.remove("--pmc")
    repeats = 5
    if "--repeats" in argv:
        i = argv.index("--repeats"); repeats = int(argv[i + 1]); del argv[i:i + 2]
    L = _capi.lib()
    for cid in [int(a) for a in argv] or [3]:
        cfg = bench.CONFIGS[cid]
        scene = bench.build_scene(cfg)
        cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(cfg["w"], cfg["h"])
        r = rt.Renderer(0); r.download = False
        st = r.GetSettings(); st.SceneMode, st.Bounces = cfg["mode"], cfg["bounces"]
        r.OnResize(cfg["w"], cfg["h"]); r.upload(scene, cam)
        r.render_only(1); r.sync()
        cap = 24_000_000
        buf = np.zeros((cap, 8), np.float32)
        n = C.c_uint32()
        _capi.check(r._h, L.rt_debug_dump_rays(r._h, 2, _capi.ptr(buf), cap, C.byref(n)), "rt_debug_dump_rays")
        total = min(n.value, cap)
        rays = buf[:total]
        bounce = rays[:, 3].view(np.uint32)
        out = {"config": cid, "workload": cfg["name"], "rays_walking": int(total), "repeats": repeats, "sets": {}}

        def walk(sub, variant, refill):
            hits = np.zeros((len(sub), 4), np.float32); ms = C.c_float()
            a = np.ascontiguousarray(sub)
            _capi.check(r._h, L.rt_debug_walk_rays(r._h, _capi.ptr(a), len(a), variant, refill, repeats, _capi.ptr(hits), C.byref(ms)), "rt_debug_walk_rays")
            return ms.value, hits

        sets = (("scattered", bounce >= 2),) if pmc else (("primary", bounce == 1), ("scattered", bounce >= 2))
        for name, mask in sets:
            sub = rays[mask]
            if not len(sub):
                continue
            order = np.argsort(sort_key(sub), kind="stable")
            rec = {"rays": int(len(sub))}
            base_hits = None
            for refill in ((1,) if pmc else (0, 1)):
                tag = "refill" if refill else "one_lane_per_ray"
                t_path, h_path = walk(sub, 0, refill)
                t_sort, h_sort = walk(sub[order], 1, refill)
                rec[tag] = {"path_order_ms": t_path, "sorted_ms": t_sort, "speedup_sorted": t_path / t_sort}
                if not pmc:
                    shuffled = np.random.default_rng(1).permutation(len(sub))
                    t_shuf, h_shuf = walk(sub[shuffled], 0, refill)
                    rec[tag]["shuffled_ms"] = t_shuf
                    back = np.empty_like(h_sort); back[order] = h_sort
                    back2 = np.empty_like(h_shuf); back2[shuffled] = h_shuf
                    rec[tag]["same_hits_in_all_orders"] = bool(np.array_equal(h_path.view(np.uint32), back.view(np.uint32)) and
                                                               np.array_equal(h_path.view(np.uint32), back2.view(np.uint32)))
                    base_hits = h_path
            if base_hits is not None:
                ids = base_hits[:, 3].view(np.int32)
                rec["hit_fraction"] = float((ids >= 0).mean()); rec["not_walked"] = int((ids == -3).sum())
            out["sets"][name] = rec
        print(json.dumps(out), flush=True)
        del r, scene


if __name__ == "__main__":
    main()
