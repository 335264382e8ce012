#!/usr/bin/env python3
"""Generates the golden fixtures in this directory FROM THE REFERENCE'S OWN OBJECTS (oracle/_ref/libref.so: the
unmodified KDTreeCPU, Intersections, Camera, Walnut::Random/pcg32 and glm, compiled by oracle/Makefile).

Run in the build container, where /root/reference exists:   python tests/golden/make_golden.py
The fixtures are data only (inputs + the reference's outputs); the reference itself does not travel to the GPU box.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import orc, ref  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
import ctypes as C  # noqa: E402


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    assert ref.available(), "build oracle/_ref first (make -C oracle)"
    L = ref.lib()
    rng = np.random.default_rng(20240611)

    # ---- RNG + camera known answers -----------------------------------------------------------------------------
    kat = dict(pcg32_default=ref.pcg32_u32(32), pcg32_seed_12345=ref.pcg32_u32(32, seed=12345),
               random_float_bits=ref.random_floats(32).view(np.uint32))
    for (w, h) in [(64, 36), (320, 180), (1280, 720), (1920, 1080), (3840, 2160)]:
        cam = ref.camera(70.0, 0.05, 100.0, w, h)
        tag = "%dx%d" % (w, h)
        for k in ("pos", "fwd", "proj", "inv_proj", "view", "inv_view"):
            kat["cam_%s_%s" % (tag, k)] = cam[k]
        d = cam["dirs"].reshape(h, w, 3)
        kat["cam_%s_dirs_fnv" % tag] = np.array([orc.fnv1a(cam["dirs"])], np.uint32)
        kat["cam_%s_corners" % tag] = np.stack([d[0, 0], d[0, w - 1], d[h // 2, w // 2], d[h - 1, w - 1]])
        kat["cam_%s_centre_column_x" % tag] = d[:, w // 2, 0].copy()
        if w == 64:
            kat["cam_%s_dirs" % tag] = cam["dirs"]
    np.savez_compressed(os.path.join(HERE, "ref_kat.npz"), **kat)

    # ---- glm primitives on random inputs --------------------------------------------------------------------------
    n = 256
    a = rng.standard_normal((n, 3)).astype(np.float32); b = rng.standard_normal((n, 3)).astype(np.float32)
    s = rng.random(n).astype(np.float32)
    out = dict(a=a, b=b, s=s, normalize=np.zeros((n, 3), np.float32), reflect=np.zeros((n, 3), np.float32),
               mix=np.zeros((n, 3), np.float32), dot=np.zeros(n, np.float32), cross=np.zeros((n, 3), np.float32))
    for i in range(n):
        L.ref_glm_normalize(p(a[i]), p(out["normalize"][i]))
        L.ref_glm_reflect(p(a[i]), p(b[i]), p(out["reflect"][i]))
        L.ref_glm_mix(p(a[i]), p(b[i]), float(s[i]), p(out["mix"][i]))
        out["dot"][i] = L.ref_glm_dot(p(a[i]), p(b[i]))
        L.ref_glm_cross(p(a[i]), p(b[i]), p(out["cross"][i]))
    m = (rng.standard_normal((64, 16)) + np.eye(4).reshape(16) * 2).astype(np.float32)
    v4 = rng.standard_normal((64, 4)).astype(np.float32)
    out["mat"] = m; out["vec4"] = v4
    out["inverse"] = np.zeros((64, 16), np.float32); out["mat_vec"] = np.zeros((64, 4), np.float32)
    for i in range(64):
        L.ref_glm_inverse(p(m[i]), p(out["inverse"][i]))
        L.ref_glm_mat_vec(p(m[i]), p(v4[i]), p(out["mat_vec"][i]))
    c = np.concatenate([rng.random((200, 3)) * 4.0, rng.random((56, 3)) * 0.01]).astype(np.float32)
    c[0] = 0.0
    out["tonemap_in"] = c; out["tonemap_glibc_powf"] = np.zeros_like(c)
    for i in range(len(c)):
        L.ref_glm_aces_srgb(p(c[i]), p(out["tonemap_glibc_powf"][i]))
    np.savez_compressed(os.path.join(HERE, "ref_glm.npz"), **out)

    # ---- Intersections::aabbIntersect / triIntersect ----------------------------------------------------------------
    n = 4096
    center = rng.uniform(-2, 2, (n, 3)).astype(np.float32); extent = rng.uniform(0, 1.5, (n, 3)).astype(np.float32)
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:512, 0] = 0.0                       # exact zero components: inverse = inf, products 0*inf = NaN
    d[256:768, 1] = 0.0
    center[:128, 0] = o[:128, 0]           # (center - origin) == 0 on the zero-direction axis -> NaN slabs
    extent[64:192, 0] = 0.0
    with np.errstate(divide="ignore"):
        inv_d = (np.float32(1.0) / d).astype(np.float32)
    hit = np.zeros(n, np.int32); tnear = np.full(n, np.float32(3.4028234663852886e38), np.float32)
    for i in range(n):
        hit[i] = L.ref_aabb(p(center[i]), p(extent[i]), p(o[i]), p(inv_d[i]), p(tnear[i:i + 1]))
    v0 = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    v1 = (v0 + rng.uniform(-1, 1, (n, 3))).astype(np.float32); v2 = (v0 + rng.uniform(-1, 1, (n, 3))).astype(np.float32)
    to = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    cen = (v0 + v1 + v2) / np.float32(3.0)
    td = (cen + rng.normal(0, 0.4, (n, 3)) - to).astype(np.float32)        # aimed near the triangle: ~half hit
    td /= np.linalg.norm(td, axis=1, keepdims=True).astype(np.float32)
    thit = np.zeros(n, np.int32); tt = np.zeros(n, np.float32); tu = np.zeros(n, np.float32); tv = np.zeros(n, np.float32)
    for i in range(n):
        thit[i] = L.ref_tri(p(to[i]), p(td[i]), p(v0[i]), p(v1[i]), p(v2[i]), p(tt[i:i + 1]), p(tu[i:i + 1]), p(tv[i:i + 1]))
    np.savez_compressed(os.path.join(HERE, "ref_intersections.npz"), center=center, extent=extent, origin=o, inv_d=inv_d,
                        aabb_hit=hit, aabb_tnear=tnear, v0=v0, v1=v1, v2=v2, tri_origin=to, tri_dir=td, tri_hit=thit,
                        tri_t=tt, tri_u=tu, tri_v=tv)

    # ---- KDTreeCPU: build + intersect on the two small meshes ----------------------------------------------------------
    w, h = 160, 90
    cam = ref.camera(70.0, 0.05, 100.0, w, h)
    for name, mesh in (("grid40", scenes.displaced_grid(40)), ("cubesphere16", scenes.cube_sphere(16))):
        verts, tris, normals9, tri_mat, mats = mesh
        kd = ref.KDTree(verts, tris)
        flat = kd.export()
        leaves = (flat["left"] < 0) & (flat["right"] < 0)
        # keep the leaves' triangle lists only (interior nodes' lists are never read by the traversal)
        leaf_first = np.zeros(len(leaves), np.int32); leaf_count = np.zeros(len(leaves), np.int32); chunks = []
        pos = 0
        for i in np.nonzero(leaves)[0]:
            f, c = flat["first"][i], flat["count"][i]
            chunks.append(flat["leaf_idx"][f:f + c]); leaf_first[i] = pos; leaf_count[i] = c; pos += c
        ids, t, u, v = kd.intersect(cam["pos"], cam["dirs"])
        # bounce-like rays: origins on the surface of earlier hits, random directions
        hitmask = ids >= 0
        o2 = (cam["pos"][None, :] + cam["dirs"][hitmask] * t[hitmask][:, None]).astype(np.float32)[:2000]
        d2 = rng.standard_normal(o2.shape).astype(np.float32)
        d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
        ids2, t2, u2, v2_ = kd.intersect(o2, d2)
        np.savez_compressed(os.path.join(HERE, "ref_kd_%s.npz" % name),
                            mesh_fnv=np.array([orc.fnv1a(verts), orc.fnv1a(tris)], np.uint32),
                            center=flat["center"], extent=flat["extent"], left=flat["left"], right=flat["right"],
                            first=leaf_first, count=leaf_count, leaf_idx=np.concatenate(chunks).astype(np.int32),
                            depth=np.array([flat["depth"]], np.int32), cam_pos=cam["pos"], width=np.array([w, h], np.int32),
                            primary_ids=ids, primary_t=t, primary_u=u, primary_v=v,
                            ray_o=o2, ray_d=d2, ray_ids=ids2, ray_t=t2, ray_u=u2, ray_v=v2_)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
