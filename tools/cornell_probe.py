#!/usr/bin/env python3
"""The reference application's own kind of scene (a Cornell-box-class OBJ through the ingest, bench.py configuration 6): which
walk order RT_ORDER_AUTO picks, ms per Render, nodes and triangles per ray, and what a forced front-to-back walk costs in rays
walked again because of near ties.   python tools/cornell_probe.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    out = {}
    for order in ("auto", "front-to-back"):
        rec, scene, cam, r = bench.run_single(6, steps, 3, order, torch, 0)
        roof = rec["roofline"]
        out[order] = {"order_in_use": roof["traversal"], "ms_per_render": rec["ms_per_step"], "Mrays_s": rec["value"],
                      "kernel_ms": roof["kernel_avg_ms"], "nodes_per_ray": roof["nodes_per_ray"], "tris_per_ray": roof["tris_per_ray"],
                      "near_tie_rewalks_per_frame": roof["near_tie_rewalks_per_frame"], "rays_per_frame": rec["rays_per_frame"],
                      "reference_order_counts": roof["reference_order_counts"], "grazing_ties": roof["grazing_ties"],
                      "ms_per_frame_x16": rec["throughput_mode"]["ms_per_frame"],
                      "camera_move_ms": rec["view_change"]["rotate"]["ms_per_render"], "kd_nodes": scene.kd_tree.getNumNodes(),
                      "triangles": int(len(scene.tri_idx))}
        del r
    print(json.dumps({"config": 6, "workload": bench.CONFIGS[6]["name"], "orders": out}))


if __name__ == "__main__":
    main()
