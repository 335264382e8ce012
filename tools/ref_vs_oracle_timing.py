#!/usr/bin/env python3
"""Build container only: the reference's own KDTreeCPU::intersect (oracle/_ref/libref.so, the reference's unmodified objects)
and the oracle's restatement of it, timed on the same primary rays of the 69,938-triangle mesh at 1920x1080, one thread each.
Says how far the CPU baseline that bench.py times (kind "port": the restatement) is from the reference's own code.

    python tools/ref_vs_oracle_timing.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle import orc, ref  # noqa: E402
from raytracer_amd import scenes  # noqa: E402


def main():
    if not ref.available():
        raise SystemExit("oracle/_ref/libref.so is missing (make -C oracle ref, where /root/reference exists)")
    w, h = 1920, 1080
    verts, tris, normals9, tri_mat, _m = scenes.displaced_grid(187, mat_offset=0)
    t0 = time.perf_counter(); kd = ref.KDTree(verts, tris); t_build = time.perf_counter() - t0
    cam = orc.camera(70.0, 0.05, 100.0, w, h)
    rows = np.arange(0, h, 4)                                    # every 4th row: 518,400 rays
    dirs = cam["dirs"].reshape(h, w, 3)[rows].reshape(-1, 3).copy()
    t0 = time.perf_counter(); ids_r, t_r, u_r, v_r = kd.intersect(cam["pos"], dirs); t_ref = time.perf_counter() - t0
    osc = orc.Scene(); osc.set_mesh(verts, tris, normals9, tri_mat); osc.import_kd(kd.export())
    t0 = time.perf_counter(); ids_o, t_o, u_o, v_o, cnt = osc.trace_primary(0, cam["pos"], dirs); t_orc = time.perf_counter() - t0
    hit = ids_r >= 0
    same = bool(np.array_equal(ids_r, ids_o) and
                np.array_equal(t_r[hit].view(np.uint32), t_o[hit].view(np.uint32)))
    print(json.dumps({"rays": len(dirs), "hits": int(hit.sum()), "reference_KDTreeCPU_intersect_s": t_ref, "oracle_restatement_s": t_orc,
                      "reference_Mrays_s": len(dirs) / t_ref / 1e6, "oracle_Mrays_s": len(dirs) / t_orc / 1e6,
                      "oracle_over_reference_time": t_orc / t_ref, "same_hits_and_t_bits": same, "reference_build_s": t_build,
                      "threads": 1, "mesh": "69,938-triangle displaced grid", "frame": "1920x1080, every 4th row"}))


if __name__ == "__main__":
    main()
