"""ctypes binding of the parity oracle (oracle/liborc.so, built from rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborc.so")

RNG_SERIAL_LIBM = 0
RNG_PIXEL_HASH = 1
RNG_SERIAL_LIBM_DOUBLE = 2
ID_TRI_BIT = 0x40000000

SPHERE_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("radius", "f4"), ("material", "u4")])
MATERIAL_DTYPE = np.dtype([("albedo", "f4", 3), ("emission", "f4", 3), ("roughness", "f4"),
                           ("transparency", "f4"), ("ior", "f4")])


class Settings(C.Structure):
    _fields_ = [("accumulate", C.c_int32), ("scene_mode", C.c_int32), ("use_ace", C.c_int32),
                ("anti_aliasing", C.c_int32), ("bounces", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("sphere_tests", C.c_uint64)]


def build(force=False):
    """Compile liborc.so (and _ref/libref.so when the reference tree is present)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float
        L.orc_scene_create.restype = vp
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_set_spheres.argtypes = [vp, vp, u32]
        L.orc_scene_set_materials.argtypes = [vp, vp, u32]
        L.orc_scene_set_mesh.argtypes = [vp, vp, u32, vp, vp, vp, u32]
        L.orc_scene_build_kd.argtypes = [vp]
        L.orc_scene_import_kd.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, u32, vp]
        L.orc_scene_kd_sizes.argtypes = [vp, vp, vp, vp]
        L.orc_scene_export_kd.argtypes = [vp] + [vp] * 7
        L.orc_camera_forward_default.argtypes = [vp]
        L.orc_camera_projection.argtypes = [f32, f32, f32, f32, f32, vp, vp]
        L.orc_camera_view.argtypes = [vp, vp, vp, vp]
        L.orc_camera_ray_dirs.argtypes = [vp, vp, u32, u32, vp]
        L.orc_render_frame.argtypes = [vp, vp, vp, vp, u32, u32, u32, C.c_int, u32, vp, vp, vp, vp, C.c_int, u32]
        L.orc_trace_primary.argtypes = [vp, C.c_int, vp, vp, u32, vp, vp, vp, vp, vp]
        L.orc_pcg32_next.argtypes = [vp]; L.orc_pcg32_next.restype = u32
        L.orc_pcg32_seed.argtypes = [C.c_uint64]; L.orc_pcg32_seed.restype = C.c_uint64
        L.orc_pixel_seed.argtypes = [u32, u32, u32]; L.orc_pixel_seed.restype = C.c_uint64
        L.orc_u32_to_float.argtypes = [u32]; L.orc_u32_to_float.restype = f32
        L.orc_sincos.argtypes = [f32, vp, vp]
        L.orc_pow_inv24.argtypes = [f32]; L.orc_pow_inv24.restype = f32
        L.orc_aabb.argtypes = [vp] * 5
        L.orc_tri.argtypes = [vp] * 8
        L.orc_refract.argtypes = [vp, vp, vp, f32, vp, vp]
        L.orc_tonemap.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_fnv1a.argtypes = [vp, C.c_uint64]; L.orc_fnv1a.restype = u32
        L.orc_glm_normalize.argtypes = [vp, vp]
        L.orc_glm_reflect.argtypes = [vp, vp, vp]
        L.orc_glm_mix.argtypes = [vp, vp, f32, vp]
        L.orc_glm_dot.argtypes = [vp, vp]; L.orc_glm_dot.restype = f32
        L.orc_glm_cross.argtypes = [vp, vp, vp]
        L.orc_glm_inverse.argtypes = [vp, vp]
        L.orc_glm_mat_vec.argtypes = [vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fnv1a(arr):
    a = np.ascontiguousarray(arr)
    return lib().orc_fnv1a(_p(a), a.nbytes)


class Scene:
    """Spheres + materials + mesh + KD tree held by the oracle."""

    def __init__(self):
        self._h = lib().orc_scene_create()
        self.n_tris = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None

    def set_spheres(self, spheres):
        s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        lib().orc_scene_set_spheres(self._h, _p(s), len(s))

    def set_materials(self, mats):
        m = np.ascontiguousarray(mats, dtype=MATERIAL_DTYPE)
        lib().orc_scene_set_materials(self._h, _p(m), len(m))

    def set_mesh(self, verts, tri_idx, normals9, tri_mat):
        v = _f32(verts).reshape(-1, 3)
        t = np.ascontiguousarray(tri_idx, dtype=np.uint32).reshape(-1, 3)
        n = _f32(normals9).reshape(-1, 9)
        m = np.ascontiguousarray(tri_mat, dtype=np.uint32).reshape(-1)
        assert len(t) == len(n) == len(m)
        lib().orc_scene_set_mesh(self._h, _p(v), len(v), _p(t), _p(n), _p(m), len(t))
        self.n_tris = len(t)

    def build_kd(self):
        rc = lib().orc_scene_build_kd(self._h)
        if rc:
            raise RuntimeError("orc_scene_build_kd failed: %d" % rc)

    def import_kd(self, flat):
        c = _f32(flat["center"]); e = _f32(flat["extent"])
        arrs = [np.ascontiguousarray(flat[k], dtype=np.int32) for k in ("left", "right", "first", "count")]
        li = np.ascontiguousarray(flat["leaf_idx"], dtype=np.int32)
        rc = lib().orc_scene_import_kd(self._h, len(arrs[0]), _p(c), _p(e), *[_p(a) for a in arrs], len(li), _p(li))
        if rc:
            raise RuntimeError("orc_scene_import_kd failed: %d" % rc)

    def export_kd(self):
        nn, ni, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().orc_scene_kd_sizes(self._h, C.byref(nn), C.byref(ni), C.byref(d))
        nn, ni = nn.value, ni.value
        out = dict(center=np.zeros((nn, 3), np.float32), extent=np.zeros((nn, 3), np.float32),
                   left=np.zeros(nn, np.int32), right=np.zeros(nn, np.int32), first=np.zeros(nn, np.int32),
                   count=np.zeros(nn, np.int32), leaf_idx=np.zeros(ni, np.int32), depth=d.value)
        lib().orc_scene_export_kd(self._h, *[_p(out[k]) for k in
                                             ("center", "extent", "left", "right", "first", "count", "leaf_idx")])
        return out

    def render_frame(self, settings, cam_pos, dirs, w, h, frame_index, accum, image, rng_mode=RNG_PIXEL_HASH,
                     seed=1337, primary_ids=None, counters=None, nthreads=1, row_step=1):
        pos = _f32(cam_pos)
        assert dirs.dtype == np.float32 and dirs.flags.c_contiguous and dirs.size == w * h * 3
        assert accum.dtype == np.float32 and accum.size == w * h * 3 and accum.flags.c_contiguous
        assert image.dtype == np.float32 and image.size == w * h * 4 and image.flags.c_contiguous
        rc = lib().orc_render_frame(self._h, C.byref(settings), _p(pos), _p(dirs), w, h, frame_index, rng_mode, seed,
                                    _p(accum), _p(image), _p(primary_ids),
                                    C.byref(counters) if counters is not None else None, nthreads, row_step)
        if rc:
            raise RuntimeError("orc_render_frame failed: %d" % rc)

    def trace_primary(self, scene_mode, cam_pos, dirs):
        pos = _f32(cam_pos)
        d = _f32(dirs).reshape(-1, 3)
        n = len(d)
        ids = np.zeros(n, np.int32); t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        cnt = Counters()
        rc = lib().orc_trace_primary(self._h, scene_mode, _p(pos), _p(d), n, _p(ids), _p(t), _p(u), _p(v), C.byref(cnt))
        if rc:
            raise RuntimeError("orc_trace_primary failed: %d" % rc)
        return ids, t, u, v, cnt


def serial_rng_reset():
    lib().orc_serial_rng_reset()


def camera(fov, znear, zfar, w, h, pos=(0.0, 1.0, 2.0), fwd=None):
    """Camera matrices + ray directions as the reference Camera computes them."""
    L = lib()
    if fwd is None:
        fwd = np.zeros(3, np.float32)
        L.orc_camera_forward_default(_p(fwd))
    fwd = _f32(fwd); pos = _f32(pos)
    proj = np.zeros(16, np.float32); iproj = np.zeros(16, np.float32)
    view = np.zeros(16, np.float32); iview = np.zeros(16, np.float32)
    L.orc_camera_projection(fov, float(w), float(h), znear, zfar, _p(proj), _p(iproj))
    L.orc_camera_view(_p(pos), _p(fwd), _p(view), _p(iview))
    dirs = np.zeros((h * w, 3), np.float32)
    L.orc_camera_ray_dirs(_p(iproj), _p(iview), w, h, _p(dirs))
    return dict(pos=pos, fwd=fwd, proj=proj, inv_proj=iproj, view=view, inv_view=iview, dirs=dirs)


def ray_dirs(inv_proj, inv_view, w, h):
    ip = _f32(inv_proj).reshape(16); iv = _f32(inv_view).reshape(16)
    dirs = np.zeros((h * w, 3), np.float32)
    lib().orc_camera_ray_dirs(_p(ip), _p(iv), w, h, _p(dirs))
    return dirs
