"""ctypes binding of oracle/_ref/libref.so: the reference's OWN objects (KDTreeCPU, Intersections, Camera,
Walnut::Random/pcg32, glm) compiled unmodified from /root/reference by oracle/Makefile.

TEST INFRASTRUCTURE ONLY.  Exists only where the reference tree exists (the build container); `available()`
is False elsewhere and the tests that use it skip.  Used to pin oracle/rt_oracle.c and to generate the golden
fixtures under tests/golden/ (tests/golden/make_golden.py).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref.so")

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
        L.ref_kd_build.argtypes = [C.c_int, vp, C.c_int, vp]; L.ref_kd_build.restype = vp
        L.ref_kd_free.argtypes = [vp]
        L.ref_kd_sizes.argtypes = [vp, vp, vp, vp]
        L.ref_kd_export.argtypes = [vp] * 8
        L.ref_kd_intersect.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
        L.ref_kd_intersect_rays.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
        L.ref_aabb.argtypes = [vp] * 5
        L.ref_tri.argtypes = [vp] * 8
        L.ref_random_floats.argtypes = [u32, vp]
        L.ref_pcg32_u32.argtypes = [u32, vp]
        L.ref_pcg32_seeded_u32.argtypes = [C.c_uint64, u32, vp]
        L.ref_camera.argtypes = [f32, f32, f32, u32, u32] + [vp] * 7
        L.ref_glm_normalize.argtypes = [vp, vp]
        L.ref_glm_reflect.argtypes = [vp, vp, vp]
        L.ref_glm_mix.argtypes = [vp, vp, f32, vp]
        L.ref_glm_dot.argtypes = [vp, vp]; L.ref_glm_dot.restype = f32
        L.ref_glm_cross.argtypes = [vp, vp, vp]
        L.ref_glm_inverse.argtypes = [vp, vp]
        L.ref_glm_mat_vec.argtypes = [vp, vp, vp]
        L.ref_glm_aces_srgb.argtypes = [vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class KDTree:
    """The reference's KDTreeCPU built on (verts, tri_idx)."""

    def __init__(self, verts, tri_idx):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tri_idx, np.uint32).reshape(-1, 3)
        self._h = lib().ref_kd_build(len(t), _p(t), len(v), _p(v))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_kd_free(self._h)
            self._h = None

    def export(self):
        nn, ni, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().ref_kd_sizes(self._h, C.byref(nn), C.byref(ni), C.byref(d))
        nn, ni = nn.value, ni.value
        out = dict(center=np.zeros((nn, 3), np.float32), extent=np.zeros((nn, 3), np.float32),
                   left=np.zeros(nn, np.int32), right=np.zeros(nn, np.int32), first=np.zeros(nn, np.int32),
                   count=np.zeros(nn, np.int32), leaf_idx=np.zeros(ni, np.int32), depth=d.value)
        lib().ref_kd_export(self._h, *[_p(out[k]) for k in
                                       ("center", "extent", "left", "right", "first", "count", "leaf_idx")])
        return out

    def intersect(self, origin, dirs):
        o = np.ascontiguousarray(origin, np.float32)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(d)
        ids = np.zeros(n, np.int32); t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        if o.size == 3:
            lib().ref_kd_intersect(self._h, _p(o), _p(d), n, _p(ids), _p(t), _p(u), _p(v))
        else:
            o = o.reshape(-1, 3); assert len(o) == n
            lib().ref_kd_intersect_rays(self._h, _p(o), _p(d), n, _p(ids), _p(t), _p(u), _p(v))
        return ids, t, u, v


def camera(fov, znear, zfar, w, h, want_dirs=True):
    pos = np.zeros(3, np.float32); fwd = np.zeros(3, np.float32)
    mats = [np.zeros(16, np.float32) for _ in range(4)]
    dirs = np.zeros((h * w, 3), np.float32) if want_dirs else None
    lib().ref_camera(fov, znear, zfar, w, h, _p(pos), _p(fwd), *[_p(m) for m in mats], _p(dirs))
    return dict(pos=pos, fwd=fwd, proj=mats[0], inv_proj=mats[1], view=mats[2], inv_view=mats[3], dirs=dirs)


def random_floats(n):
    out = np.zeros(n, np.float32)
    lib().ref_random_floats(n, _p(out))
    return out


def pcg32_u32(n, seed=None):
    out = np.zeros(n, np.uint32)
    if seed is None:
        lib().ref_pcg32_u32(n, _p(out))
    else:
        lib().ref_pcg32_seeded_u32(seed, n, _p(out))
    return out
