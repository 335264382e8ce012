"""ctypes binding of the C ABI declared in include/rt_mi355x.h (raytracer_amd/lib/librt_mi355x.so).

There is no CPU fallback: if the library is missing or does not load, importing this module's `lib()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_MI355X_LIB=<path> loads another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(_HERE, "lib", "librt_mi355x.so")

RT_OK = 0
SCENE_MESH, SCENE_SPHERES, SCENE_BOTH = 0, 1, 2
ORDER_AUTO, ORDER_REFERENCE, ORDER_FRONT_TO_BACK = 0, 1, 2
ID_TRI_BIT = 0x40000000

# every symbol include/rt_mi355x.h declares
EXPORTS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_resize", "rt_set_settings", "rt_set_spheres",
    "rt_set_materials", "rt_set_mesh", "rt_set_kdtree", "rt_traversal_in_use", "rt_set_camera", "rt_set_ray_dirs", "rt_set_bands", "rt_render",
    "rt_render_frames", "rt_render_spheres_noalloc", "rt_render_spheres_noalloc_setup", "rt_resolve", "rt_reset_accum", "rt_sync", "rt_read_image", "rt_map_image", "rt_map_image_pipelined", "rt_read_accum", "rt_set_record_ids",
    "rt_read_primary_ids", "rt_read_ray_dirs", "rt_set_stats", "rt_read_counters", "rt_reset_counters", "rt_read_wave_clock",
    "rt_get_kernel_time", "rt_reset_kernel_time", "rt_set_kernel_timing", "rt_device_accum", "rt_device_image",
    "rt_debug_dump_rays", "rt_debug_walk_rays",
    "rt_group_create", "rt_group_destroy", "rt_group_size", "rt_group_ctx", "rt_group_last_error", "rt_group_render_tiles",
    "rt_group_render_samples", "rt_group_read_accum", "rt_group_issue_overlap",
    "rt_host_camera_default_forward", "rt_host_camera_matrices", "rt_host_kd_build", "rt_host_kd_flat", "rt_host_kd_free",
]


class RtSettings(C.Structure):
    _fields_ = [("accumulate", C.c_int32), ("scene_mode", C.c_int32), ("use_ace", C.c_int32),
                ("anti_aliasing", C.c_int32), ("bounces", C.c_uint32), ("seed", C.c_uint32), ("traversal", C.c_int32),
                ("reuse_first_hits", C.c_int32)]


class RtKdFlat(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("center", C.c_void_p), ("extent", C.c_void_p), ("left", C.c_void_p),
                ("right", C.c_void_p), ("first", C.c_void_p), ("count", C.c_void_p), ("n_leaf_idx", C.c_uint32),
                ("leaf_idx", C.c_void_p)]


class RtCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("prepass_nodes", C.c_uint64), ("prepass_tris", C.c_uint64), ("near_ties", C.c_uint64),
                ("primary_hits_reused", C.c_uint64), ("empty_unit_rays", C.c_uint64),
                ("view_setups", C.c_uint64), ("entry_nodes_spared", C.c_uint64), ("entry_tris_spared", C.c_uint64)]


class RtKernelTime(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("total_ms", C.c_double), ("last_ms", C.c_double), ("prepass_total_ms", C.c_double),
                ("frames", C.c_uint64)]


_lib = None


def _preload_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as the system one).  A process must hold ONE HIP
    runtime, so when torch is installed its copy is loaded first -- by path, without importing torch -- and this
    library's libamdhip64.so.7 dependency then resolves to it; a later `import torch` reuses the same file."""
    import importlib.util
    import sys
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "raytracer_amd: %s is missing -- build it with `make -C raytracer_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, f32, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_float, C.c_size_t
    sig = {
        "rt_create": ([C.c_int], vp), "rt_destroy": ([vp], None), "rt_last_error": ([vp], C.c_char_p),
        "rt_set_stream": ([vp, vp], C.c_int), "rt_resize": ([vp, u32, u32], C.c_int),
        "rt_set_settings": ([vp, vp], C.c_int), "rt_set_spheres": ([vp, vp, u32], C.c_int),
        "rt_set_materials": ([vp, vp, u32], C.c_int), "rt_set_mesh": ([vp, vp, u32, vp, vp, vp, u32], C.c_int),
        "rt_set_kdtree": ([vp, vp], C.c_int), "rt_traversal_in_use": ([vp], C.c_int), "rt_set_camera": ([vp, vp, vp, vp], C.c_int),
        "rt_set_ray_dirs": ([vp, vp, sz], C.c_int), "rt_set_bands": ([vp, u32, u32], C.c_int),
        "rt_render": ([vp, u32], C.c_int), "rt_render_frames": ([vp, u32, u32], C.c_int),
        "rt_render_spheres_noalloc": ([C.c_int, vp, vp, u32, vp, u32, vp, vp, vp, vp, u32, u32, u32, vp, vp], C.c_int),
        "rt_render_spheres_noalloc_setup": ([C.c_int], C.c_int),
        "rt_resolve": ([vp, u32], C.c_int), "rt_reset_accum": ([vp], C.c_int), "rt_sync": ([vp], C.c_int), "rt_read_image": ([vp, vp], C.c_int), "rt_map_image": ([vp], vp), "rt_map_image_pipelined": ([vp], vp),
        "rt_read_accum": ([vp, vp], C.c_int), "rt_set_record_ids": ([vp, C.c_int], C.c_int),
        "rt_read_primary_ids": ([vp, vp], C.c_int), "rt_read_ray_dirs": ([vp, vp], C.c_int),
        "rt_set_stats": ([vp, C.c_int], C.c_int), "rt_read_counters": ([vp, vp], C.c_int),
        "rt_reset_counters": ([vp], C.c_int), "rt_set_kernel_timing": ([vp, u32], C.c_int), "rt_read_wave_clock": ([vp, vp, u32, vp], C.c_int), "rt_get_kernel_time": ([vp, vp], C.c_int),
        "rt_reset_kernel_time": ([vp], C.c_int), "rt_device_accum": ([vp], vp), "rt_device_image": ([vp], vp),
        "rt_debug_dump_rays": ([vp, u32, vp, u32, vp], C.c_int),
        "rt_debug_walk_rays": ([vp, vp, u32, C.c_int, C.c_int, u32, vp, vp], C.c_int),
        "rt_group_create": ([vp, C.c_int], vp), "rt_group_destroy": ([vp], None), "rt_group_size": ([vp], C.c_int),
        "rt_group_ctx": ([vp, C.c_int], vp), "rt_group_last_error": ([vp], C.c_char_p),
        "rt_group_render_tiles": ([vp, u32, u32], C.c_int), "rt_group_render_samples": ([vp, u32, u32], C.c_int),
        "rt_group_read_accum": ([vp, vp], C.c_int), "rt_group_issue_overlap": ([vp], C.c_int),
        "rt_host_camera_default_forward": ([vp], None),
        "rt_host_camera_matrices": ([f32, f32, f32, u32, u32] + [vp] * 6, None),
        "rt_host_kd_build": ([vp, u32, vp, u32, u32, C.c_int], vp),
        "rt_host_kd_flat": ([vp, vp, vp], C.c_int), "rt_host_kd_free": ([vp], None),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class RtError(RuntimeError):
    pass


def check(ctx_handle, rc, what):
    if rc != RT_OK:
        msg = lib().rt_last_error(ctx_handle)
        raise RtError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
