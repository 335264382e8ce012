"""Scene data for the tracing path: the reference application's default sphere scene and the synthetic
bounded-overlap meshes the benchmark configurations use.

Data only (inputs to the path), as numpy arrays in the layouts the C ABI takes:
  spheres   : SPHERE_DTYPE   {x, y, z, radius, material}                      (Scene.h:10-15)
  materials : MATERIAL_DTYPE {albedo[3], emission[3], roughness, transparency, ior} (Scene.h:26-35)
  mesh      : verts (V,3) f32, tri_idx (T,3) u32 (indexed, shared vertices), normals (T,9) f32 (three per-corner
              normals per triangle, Scene.h:17-24), tri_mat (T,) u32
"""
import numpy as np

SPHERE_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("radius", "f4"), ("material", "u4")])
MATERIAL_DTYPE = np.dtype([("albedo", "f4", 3), ("emission", "f4", 3), ("roughness", "f4"),
                           ("transparency", "f4"), ("ior", "f4")])


def make_material(albedo=(1.0, 1.0, 1.0), emission=(0.0, 0.0, 0.0), roughness=1.0, transparency=0.0, ior=1.0):
    m = np.zeros((), MATERIAL_DTYPE)
    m["albedo"] = albedo; m["emission"] = emission
    m["roughness"] = roughness; m["transparency"] = transparency; m["ior"] = ior
    return m


def default_materials(mat_offset=0):
    """The seven hard-coded materials of the reference application (RaytracerApp.cpp:136-187)."""
    mats = [
        make_material((0.8, 0.8, 0.8), roughness=1.0),                       # floor / walls
        make_material((0.35, 1.0, 0.17), roughness=1.0),                     # left wall
        make_material((1.0, 0.0, 0.0), roughness=1.0),                       # right wall
        make_material((0.0, 0.5, 1.0), roughness=0.9),                       # right sphere
        make_material((1.0, 0.8, 0.0), roughness=0.0, transparency=1.0, ior=1.1),  # left (glass) sphere
        make_material((1.0, 1.0, 1.0), emission=(5.0, 5.0, 5.0)),            # light
        make_material((1.0, 1.0, 1.0), roughness=0.0),                       # mirror sphere
    ]
    return np.array(mats, dtype=MATERIAL_DTYPE)


def default_spheres(mat_offset=0):
    """The nine spheres of the reference application (RaytracerApp.cpp:191-254): five r=1000 walls, three
    small spheres and the r=1 light."""
    rows = [
        (0.0, 1002.0, 0.0, 1000.0, 0), (0.0, -1000.0, 0.0, 1000.0, 0), (0.0, 0.0, -1002.0, 1000.0, 0),
        (-1002.0, 0.0, 0.0, 1000.0, 1), (1002.0, 0.0, 0.0, 1000.0, 2),
        (0.5, 0.3, -0.3, 0.3, 3), (-0.5, 0.3, -0.1, 0.3, 4), (0.0, 0.5, -1.5, 0.5, 6),
        (0.0, 2.9, 0.0, 1.0, 5),
    ]
    s = np.zeros(len(rows), SPHERE_DTYPE)
    for i, r in enumerate(rows):
        s[i] = (np.float32(r[0]), np.float32(r[1]), np.float32(r[2]), np.float32(r[3]), r[4] + mat_offset)
    return s


def _grid_tris(nx, nz):
    """Two triangles per quad over an (nx+1) x (nz+1) vertex lattice, row-major in z."""
    i, k = np.meshgrid(np.arange(nx), np.arange(nz), indexing="xy")
    v00 = (k * (nx + 1) + i).ravel()
    v10 = v00 + 1
    v01 = v00 + (nx + 1)
    v11 = v01 + 1
    t = np.empty((nx * nz, 2, 3), np.uint32)
    t[:, 0] = np.stack([v00, v01, v10], 1)
    t[:, 1] = np.stack([v10, v01, v11], 1)
    return t.reshape(-1, 3)


def displaced_grid(n=187, drop=0, x_range=(-1.5, 1.5), z_range=(-3.0, 0.0), emissive_period=9, mat_offset=0):
    """Displaced height-field: n x n quads (2 n^2 triangles) over x in x_range, z in z_range with
    y = 0.3 + 0.15 sin(4x) cos(5z); shared (indexed) vertices, analytic per-vertex normals.  Valence <= 6 everywhere,
    so no region is covered by more than a handful of triangle boxes (the reference KD builder only terminates on
    such meshes).  Every `emissive_period`-th quad row is emissive so the path-traced image is not black (the
    reference has no ambient light).  Returns (verts, tri_idx, normals9, tri_mat, materials)."""
    xs = np.linspace(x_range[0], x_range[1], n + 1, dtype=np.float64)
    zs = np.linspace(z_range[0], z_range[1], n + 1, dtype=np.float64)
    X, Z = np.meshgrid(xs, zs, indexing="xy")
    Y = 0.3 + 0.15 * np.sin(4.0 * X) * np.cos(5.0 * Z)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    dydx = 0.6 * np.cos(4.0 * X) * np.cos(5.0 * Z)
    dydz = -0.75 * np.sin(4.0 * X) * np.sin(5.0 * Z)
    nrm = np.stack([-dydx, np.ones_like(X), -dydz], -1).reshape(-1, 3)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    tri = _grid_tris(n, n)
    if drop:
        tri = tri[:-drop]
    normals9 = nrm[tri].reshape(-1, 9)
    quad_row = (np.arange(len(tri)) // 2) // n
    tri_mat = np.where(quad_row % emissive_period == emissive_period // 2, 1, 0).astype(np.uint32) + mat_offset
    mats = np.array([make_material((0.75, 0.7, 0.6), roughness=0.9),
                     make_material((1.0, 0.95, 0.9), emission=(6.0, 5.5, 5.0), roughness=1.0)], dtype=MATERIAL_DTYPE)
    return verts, tri, normals9, tri_mat, mats


def cube_sphere(n=76, center=(0.0, 0.9, -1.0), radius=0.8, mat_offset=0):
    """Cube-sphere: each cube face split into n x n quads, vertices pushed to the sphere (12 n^2 triangles),
    radial normals; faces do not share vertices along cube edges (valence stays <= 6).  The +y face is emissive.
    Returns (verts, tri_idx, normals9, tri_mat, materials)."""
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)),
             ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, -1, 0), (1, 0, 0), (0, 0, 1)),
             ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    lin = np.linspace(-1.0, 1.0, n + 1, dtype=np.float64)
    A, B = np.meshgrid(lin, lin, indexing="xy")
    base = _grid_tris(n, n)
    vs, ns, ts, ms = [], [], [], []
    for f, (w, a, b) in enumerate(faces):
        p = (np.asarray(w, np.float64)[None, None, :] + A[..., None] * np.asarray(a, np.float64)
             + B[..., None] * np.asarray(b, np.float64)).reshape(-1, 3)
        d = p / np.linalg.norm(p, axis=1, keepdims=True)
        vs.append((np.asarray(center, np.float64) + radius * d).astype(np.float32))
        ns.append(d.astype(np.float32))
        ts.append(base + np.uint32(f * (n + 1) * (n + 1)))
        ms.append(np.full(len(base), 1 if f == 2 else 0, np.uint32))
    verts = np.concatenate(vs); nrm = np.concatenate(ns); tri = np.concatenate(ts).astype(np.uint32)
    tri_mat = np.concatenate(ms) + np.uint32(mat_offset)
    normals9 = nrm[tri].reshape(-1, 9)
    mats = np.array([make_material((0.7, 0.75, 0.8), roughness=0.6),
                     make_material((1.0, 1.0, 1.0), emission=(4.0, 4.0, 4.0), roughness=1.0)], dtype=MATERIAL_DTYPE)
    return verts, tri, normals9, tri_mat, mats


def write_cornell_class_obj(directory, water_quads=50):
    """A Cornell-box-class scene as OBJ + MTL text, the kind of file the reference application loads (RaytracerApp.cpp:27-131; the
    file itself is not part of the reference tree): an open box of five quad walls, a quad light under the ceiling, a short and a
    tall block (34 large triangles in all) and a rippled surface of water_quads x water_quads quads (5,000 triangles at 50) in the
    material named "water", which the application makes transparent (RaytracerApp.cpp:48-49).  The box spans x in [-1, 1],
    y in [0, 2], z in [-2, 0.4], in front of the application's default camera.  Returns the path of the .obj file."""
    import os
    os.makedirs(directory, exist_ok=True)
    mtl = ["newmtl white", "Kd 0.73 0.73 0.73", "newmtl red", "Kd 0.65 0.05 0.05", "newmtl green", "Kd 0.12 0.45 0.15",
           "newmtl light", "Kd 0.78 0.78 0.78", "Ke 8.5 6.0 2.0", "newmtl water", "Kd 0.85 0.95 1.0", "Ni 1.33", "Ns 1024"]
    with open(os.path.join(directory, "cornell_class.mtl"), "w") as fh:
        fh.write("\n".join(mtl) + "\n")
    lines, nv = ["mtllib cornell_class.mtl"], [0]

    def quad(mat, a, b, c, d, n):
        lines.append("usemtl %s" % mat)
        for q in (a, b, c, d):
            lines.append("v %.6f %.6f %.6f" % q)
        lines.append("vn %.6f %.6f %.6f" % n)
        k = nv[0]
        lines.append("f %d//%d %d//%d %d//%d %d//%d" % (k + 1, -1, k + 2, -1, k + 3, -1, k + 4, -1))
        nv[0] += 4

    x0, x1, y0, y1, z0, z1 = -1.0, 1.0, 0.0, 2.0, -2.0, 0.4
    quad("white", (x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), (0, 1, 0))            # floor
    quad("white", (x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, -1, 0))           # ceiling
    quad("white", (x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, 1))            # back
    quad("red", (x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (1, 0, 0))              # left
    quad("green", (x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (-1, 0, 0))           # right
    quad("light", (-0.3, 1.98, -1.1), (0.3, 1.98, -1.1), (0.3, 1.98, -0.5), (-0.3, 1.98, -0.5), (0, -1, 0))

    def block(cx, cz, half, height, turn):
        import math
        c, s_ = math.cos(turn), math.sin(turn)
        corner = lambda dx, dz, y: (cx + c * dx * half - s_ * dz * half, y, cz + s_ * dx * half + c * dz * half)
        ring = [(-1, -1), (1, -1), (1, 1), (-1, 1)]
        for i in range(4):
            (ax, az), (bx, bz) = ring[i], ring[(i + 1) % 4]
            nx, nz = (ax + bx) / 2.0, (az + bz) / 2.0
            quad("white", corner(ax, az, 0.55), corner(bx, bz, 0.55), corner(bx, bz, 0.55 + height), corner(ax, az, 0.55 + height),
                 (c * nx - s_ * nz, 0.0, s_ * nx + c * nz))
        quad("white", corner(-1, -1, 0.55 + height), corner(1, -1, 0.55 + height), corner(1, 1, 0.55 + height), corner(-1, 1, 0.55 + height), (0, 1, 0))

    # (the blocks stand in the water: their sides start above the floor, so no more than a few triangle boxes meet anywhere)
    block(0.35, -0.6, 0.3, 0.25, -0.3)
    block(-0.35, -1.3, 0.3, 0.9, 0.3)
    # the water: a rippled height field just above the floor, wall to wall
    n = int(water_quads)
    xs = np.linspace(x0 + 0.01, x1 - 0.01, n + 1); zs = np.linspace(z0 + 0.01, z1 - 0.01, n + 1)
    lines.append("usemtl water")
    first = nv[0]
    for k in range(n + 1):
        for i in range(n + 1):
            x, z = xs[i], zs[k]
            y = 0.4 + 0.03 * np.sin(9.0 * x) * np.cos(7.0 * z)
            dx, dz = 0.27 * np.cos(9.0 * x) * np.cos(7.0 * z), -0.21 * np.sin(9.0 * x) * np.sin(7.0 * z)
            ln = np.sqrt(dx * dx + 1.0 + dz * dz)
            lines.append("v %.6f %.6f %.6f" % (x, y, z))
            lines.append("vn %.6f %.6f %.6f" % (-dx / ln, 1.0 / ln, -dz / ln))
    nn_first = 6 + 10                                                          # normals written so far: one per quad above
    for k in range(n):
        for i in range(n):
            a = first + k * (n + 1) + i + 1
            b, c_, d = a + 1, a + (n + 1) + 1, a + (n + 1)
            na, nb, nc, nd = (nn_first + (v - first) for v in (a, b, c_, d))
            lines.append("f %d//%d %d//%d %d//%d" % (a, na, d, nd, b, nb))
            lines.append("f %d//%d %d//%d %d//%d" % (b, nb, d, nd, c_, nc))
    path = os.path.join(directory, "cornell_class.obj")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return path
