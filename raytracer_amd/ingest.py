"""OBJ/MTL ingest into the path's mesh arrays, with the conventions of the reference application's scene set-up
(RayTracer/src/RaytracerApp.cpp:27-131, which drives tinyobjloader with `triangulate = true`):

  * every triangle gets three vertices of its own (per-face, unindexed: RaytracerApp.cpp:72-82,111-113), in face order
    over all shapes, so triangle i uses vertices 3i, 3i+1, 3i+2;
  * corner normals are glm::normalize(vn) (RaytracerApp.cpp:88-94); a corner without a normal gets (0, 1, 0), the value the
    reference initialises but then fails to store (it leaves Triangle::Normals short, which ClosestHitTriangle would read
    out of bounds);
  * a face without `usemtl` takes material 0 (std::max(MatID, 0), RaytracerApp.cpp:118);
  * MTL -> Material (RaytracerApp.cpp:38-51): Albedo = max(Kd, Ks) per channel, Emission = 2 * Ke,
    Roughness = (1024 - Ns) / 1024, IOR = Ni, Transparency = 1 for the material named "water", 0 otherwise; tinyobjloader's
    defaults for missing statements are Kd = Ks = Ke = 0, Ns = 1, Ni = 1 (tiny_obj_loader.h:1358-1392);
  * quads are cut along the shorter diagonal, compared in single precision: [0,1,2],[0,2,3] when |v2-v0|^2 < |v3-v1|^2, else
    [0,1,3],[1,2,3] (tiny_obj_loader.h:1488-1583); polygons with more than four corners are ear-clipped as the loader's built-in
    triangulation does (tiny_obj_loader.h:1714-1935, the application does not enable the earcut alternative): same projection
    axes, same ear test, same order of the emitted triangles.  tests/golden/ref_obj_polygons.npz holds what the reference's own
    loader returns for tests/golden/polygons.obj.

The application appends its own seven materials after the file's (RaytracerApp.cpp:132-187): `load_obj(..., app_materials=True)`.

Data conversion only: text in, numpy arrays out (what Scene.set_triangles / rt_set_mesh take).  The KD build guard against
meshes on which the reference builder does not terminate is `Scene.set_triangles(..., max_nodes=...)`.
"""
import os

import numpy as np

from .scenes import MATERIAL_DTYPE, default_materials, make_material

f32 = np.float32


def _parse_mtl(path):
    mats, cur = [], None
    with open(path) as fh:
        for line in fh:
            tok = line.split("#", 1)[0].split()
            if not tok:
                continue
            key = tok[0]
            if key == "newmtl":
                cur = dict(name=" ".join(tok[1:]), Kd=(0.0, 0.0, 0.0), Ks=(0.0, 0.0, 0.0), Ke=(0.0, 0.0, 0.0), Ns=1.0, Ni=1.0)
                mats.append(cur)
            elif cur is None:
                continue
            elif key in ("Kd", "Ks", "Ke") and len(tok) >= 4:
                cur[key] = tuple(float(x) for x in tok[1:4])
            elif key in ("Ns", "Ni") and len(tok) >= 2:
                cur[key] = float(tok[1])
    return mats


def materials_from_mtl(mtl):
    """RaytracerApp.cpp:38-51 in single precision."""
    out = np.zeros(len(mtl), MATERIAL_DTYPE)
    for i, m in enumerate(mtl):
        kd, ks, ke = (np.array(m[k], f32) for k in ("Kd", "Ks", "Ke"))
        albedo = np.where(kd < ks, ks, kd)                                  # glm::max(x, y) = (x < y) ? y : x
        out[i] = make_material(albedo, f32(2.0) * ke, roughness=(f32(1024.0) - f32(m["Ns"])) / f32(1024.0),
                               transparency=1.0 if m["name"] == "water" else 0.0, ior=f32(m["Ni"]))
    return out


def _index(tok, n):
    i = int(tok)
    return i - 1 if i > 0 else n + i            # OBJ indices are 1-based; negative = relative to the end


def _pnpoly(vx, vy, tx, ty):
    """Point-in-triangle parity test of the loader (tiny_obj_loader.h:1411-1422), single precision."""
    c = False
    j = 2
    for i in range(3):
        if (vy[i] > ty) != (vy[j] > ty) and tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]:
            c = not c
        j = i
    return c


def _ear_clip(face_pos):
    """The loader's built-in triangulation of a polygon with more than four corners (tiny_obj_loader.h:1714-1935), in single
    precision: corner triples (indices into the face) in the order the loader emits them."""
    n = len(face_pos)
    P = [np.array(q, f32) for q in face_pos]
    axes = [1, 2]
    eps = np.finfo(f32).eps
    for k in range(n):                                                      # the first real corner decides the projection plane
        v0, v1, v2 = P[k % n], P[(k + 1) % n], P[(k + 2) % n]
        e0, e1 = v1 - v0, v2 - v1
        cx = abs(e0[1] * e1[2] - e0[2] * e1[1]); cy = abs(e0[2] * e1[0] - e0[0] * e1[2]); cz = abs(e0[0] * e1[1] - e0[1] * e1[0])
        if cx > eps or cy > eps or cz > eps:
            if not (cx > cy and cx > cz):
                axes[0] = 0
                if cz > cx and cz > cy:
                    axes[1] = 1
            break
    remaining = list(range(n))
    out = []
    guess, iterations, previous = 0, n, n
    while len(remaining) > 3 and iterations > 0:
        m = len(remaining)
        if guess >= m:
            guess -= m
        if previous != m:
            previous, iterations = m, m
        else:
            iterations -= 1
        ind = [remaining[(guess + k) % m] for k in range(3)]
        vx = [P[i][axes[0]] for i in ind]; vy = [P[i][axes[1]] for i in ind]
        e0x, e0y, e1x, e1y = vx[1] - vx[0], vy[1] - vy[0], vx[2] - vx[1], vy[2] - vy[1]
        cross = e0x * e1y - e0y * e1x
        area = (vx[0] * vy[1] - vy[0] * vx[1]) * f32(0.5)
        if cross * area < f32(0.0):                                         # "an internal angle"
            guess += 1
            continue
        overlap = False
        for other in range(3, m):
            q = P[remaining[(guess + other) % m]]
            if _pnpoly(vx, vy, q[axes[0]], q[axes[1]]):
                overlap = True
                break
        if overlap:
            guess += 1
            continue
        out.append(tuple(ind))                                              # an ear: its middle corner leaves the polygon
        del remaining[(guess + 1) % m]
    if len(remaining) == 3:
        out.append(tuple(remaining))
    return out


def load_obj(path, app_materials=False):
    """Returns (verts (3T,3) f32, tri_idx (T,3) u32, normals9 (T,9) f32, tri_mat (T,) u32, materials, names)."""
    positions, normals, mtl, names = [], [], [], {}
    corners, face_mat = [], []                  # per triangle: three (v, vn) index pairs; material id
    cur_mat = -1
    base = os.path.dirname(os.path.abspath(path))
    with open(path) as fh:
        for line in fh:
            tok = line.split("#", 1)[0].split()
            if not tok:
                continue
            key = tok[0]
            if key == "v":
                positions.append([float(x) for x in tok[1:4]])
            elif key == "vn":
                normals.append([float(x) for x in tok[1:4]])
            elif key == "mtllib":
                for name in tok[1:]:
                    p = os.path.join(base, name)
                    if os.path.exists(p):
                        for m in _parse_mtl(p):
                            names.setdefault(m["name"], len(mtl))
                            mtl.append(m)
            elif key == "usemtl":
                cur_mat = names.get(" ".join(tok[1:]), -1)
            elif key == "f":
                face = []
                for c in tok[1:]:
                    parts = c.split("/")
                    vi = _index(parts[0], len(positions))
                    ni = _index(parts[2], len(normals)) if len(parts) > 2 and parts[2] else -1
                    face.append((vi, ni))
                if len(face) < 3:
                    continue
                if len(face) == 3:
                    tris = [(0, 1, 2)]
                elif len(face) == 4:
                    p = np.array([positions[v] for v, _ in face], f32)
                    e02, e13 = p[2] - p[0], p[3] - p[1]
                    sqr02 = (e02[0] * e02[0] + e02[1] * e02[1]) + e02[2] * e02[2]
                    sqr13 = (e13[0] * e13[0] + e13[1] * e13[1]) + e13[2] * e13[2]
                    tris = [(0, 1, 2), (0, 2, 3)] if sqr02 < sqr13 else [(0, 1, 3), (1, 2, 3)]
                else:
                    with np.errstate(all="ignore"):
                        tris = _ear_clip([positions[v] for v, _ in face])
                for t in tris:
                    corners.append([face[k] for k in t])
                    face_mat.append(max(cur_mat, 0))
    if not corners:
        raise ValueError("%s: no faces" % path)
    pos = np.array(positions, f32)
    nrm = np.array(normals, f32).reshape(-1, 3)
    T = len(corners)
    vi = np.array([[c[0] for c in tri] for tri in corners], np.int64)
    ni = np.array([[c[1] for c in tri] for tri in corners], np.int64)
    if vi.min() < 0 or vi.max() >= len(pos):
        raise ValueError("%s: vertex index out of range" % path)
    if ni.max() >= len(nrm):
        raise ValueError("%s: normal index out of range" % path)
    verts = pos[vi.reshape(-1)]                                             # 3 vertices of its own per triangle
    tri_idx = np.arange(3 * T, dtype=np.uint32).reshape(T, 3)
    n = np.tile(np.array([0.0, 1.0, 0.0], f32), (3 * T, 1))
    has = ni.reshape(-1) >= 0
    if has.any():
        raw = nrm[ni.reshape(-1)[has]]
        dot = (raw[:, 0] * raw[:, 0] + raw[:, 1] * raw[:, 1]) + raw[:, 2] * raw[:, 2]          # glm::dot, then v * (1 / sqrt)
        n[has] = raw * (f32(1.0) / np.sqrt(dot, dtype=f32))[:, None]
    normals9 = n.reshape(T, 9)
    tri_mat = np.array(face_mat, np.uint32)
    materials = materials_from_mtl(mtl)
    if len(materials) == 0:
        materials = np.array([make_material()], MATERIAL_DTYPE)            # every triangle names material 0
    if app_materials:
        materials = np.concatenate([materials, default_materials()])
    return verts, tri_idx, normals9, tri_mat, materials, [m["name"] for m in mtl]
