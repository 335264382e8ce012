"""Host-side mirror of the reference's Renderer / Scene / Camera interface for the tracing path, over the C ABI.

Same member names and meaning as RayTracer/src/Renderer.h:101-122, Scene.h:10-44 and Camera.h:6-48, so a caller of the
reference finds what it expects:

    scene, camera, renderer = Scene(), Camera(70.0, 0.05, 100.0), Renderer()
    renderer.OnResize(w, h); camera.OnResize(w, h)
    renderer.Render(scene, camera)          # one frame; GetImage() then holds the RGBA32F frame
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import RtCounters, RtKdFlat, RtKernelTime, RtSettings, check, lib, ptr
from .scenes import MATERIAL_DTYPE, SPHERE_DTYPE


class Settings:
    """Renderer::Settings (Renderer.h:103-110).  `SceneMode` (None = follow UseSphereScene) selects the added
    spheres+mesh mode; `Seed` seeds the per-(pixel, frame) random streams; `TraversalOrder` is the KD tree walk
    order (_capi.ORDER_AUTO / ORDER_REFERENCE / ORDER_FRONT_TO_BACK, see include/rt_mi355x.h); `ReuseFirstHits` makes mesh scenes
    keep each pixel's primary hit while camera and tree stay as they are (rt_settings.reuse_first_hits: same frames, bit for bit)."""

    def __init__(self):
        self.Render = True
        self.Accumulate = True
        self.UseSphereScene = False
        self.UseACE_Color = True
        self.AntiAliasing = False
        self.Bounces = 8
        self.SceneMode = None
        self.Seed = 1337
        self.TraversalOrder = _capi.ORDER_AUTO
        self.ReuseFirstHits = False

    def scene_mode(self):
        if self.SceneMode is not None:
            return int(self.SceneMode)
        return _capi.SCENE_SPHERES if self.UseSphereScene else _capi.SCENE_MESH

    def to_c(self):
        return RtSettings(int(bool(self.Accumulate)), self.scene_mode(), int(bool(self.UseACE_Color)),
                          int(bool(self.AntiAliasing)), int(self.Bounces), int(self.Seed), int(self.TraversalOrder),
                          int(bool(self.ReuseFirstHits)))


class KDTree:
    """Host KD tree with the reference builder's semantics (KDAccel/KDTreeCPU.cpp:17-52,424-632), flattened for
    upload.  Mirrors `KDTreeCPU(num_tris, tris, num_verts, verts)`."""

    def __init__(self, num_tris, tris, num_verts, verts, max_nodes=0, nthreads=8):
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        assert len(self.tris) == num_tris and len(self.verts) == num_verts
        self._free = lib().rt_host_kd_free          # (kept here: module globals may be gone when __del__ runs at interpreter exit)
        self._h = lib().rt_host_kd_build(ptr(self.verts), num_verts, ptr(self.tris), num_tris, max_nodes, nthreads)
        if not self._h:
            raise _capi.RtError("rt_host_kd_build failed (bad indices, or the node guard of %d was hit)" % max_nodes)
        self.flat = RtKdFlat()
        depth = C.c_uint32()
        lib().rt_host_kd_flat(self._h, C.byref(self.flat), C.byref(depth))
        self.depth = depth.value

    def __del__(self):
        if getattr(self, "_h", None):
            self._free(self._h)
            self._h = None

    def getNumNodes(self):
        return self.flat.n_nodes

    def arrays(self):
        """The flat tree as numpy arrays (copies)."""
        f = self.flat
        n, m = f.n_nodes, f.n_leaf_idx

        def arr(p, count, ct):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(count,)).copy() if count else np.zeros(0, ct)

        return dict(center=arr(f.center, 3 * n, C.c_float).reshape(n, 3), extent=arr(f.extent, 3 * n, C.c_float).reshape(n, 3),
                    left=arr(f.left, n, C.c_int32), right=arr(f.right, n, C.c_int32), first=arr(f.first, n, C.c_int32),
                    count=arr(f.count, n, C.c_int32), leaf_idx=arr(f.leaf_idx, m, C.c_int32), depth=self.depth)


class FlatKDTree:
    """A KD tree given directly as flat arrays (e.g. emitted by a walk of the reference's KDTreeCPU)."""

    def __init__(self, center, extent, left, right, first, count, leaf_idx, depth=0, **_):
        self._keep = [np.ascontiguousarray(center, np.float32), np.ascontiguousarray(extent, np.float32)] + \
                     [np.ascontiguousarray(a, np.int32) for a in (left, right, first, count, leaf_idx)]
        k = self._keep
        self.flat = RtKdFlat(len(k[2]), ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(k[3]), ptr(k[4]), ptr(k[5]), len(k[6]), ptr(k[6]))
        self.depth = depth


class Scene:
    """Scene (Scene.h:37-44): spheres, materials, triangles (as arrays) and the KD tree."""

    def __init__(self):
        self.spheres = np.zeros(0, SPHERE_DTYPE)
        self.materials = np.zeros(0, MATERIAL_DTYPE)
        self.verts = None
        self.tri_idx = None
        self.normals9 = None
        self.tri_mat = None
        self.kd_tree = None

    def set_triangles(self, verts, tri_idx, normals9, tri_mat, build_kd=True, **kd_args):
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        self.tri_idx = np.ascontiguousarray(tri_idx, np.uint32).reshape(-1, 3)
        self.normals9 = np.ascontiguousarray(normals9, np.float32).reshape(-1, 9)
        self.tri_mat = np.ascontiguousarray(tri_mat, np.uint32).reshape(-1)
        if build_kd:
            self.kd_tree = KDTree(len(self.tri_idx), self.tri_idx, len(self.verts), self.verts, **kd_args)


class Camera:
    """Camera (Camera.h:6-48, Camera.cpp:11-19,105-132): position (0,1,2), forward normalize(0,-0.1,-1); matrices
    recomputed on resize.  Ray directions are generated on the device from the inverse matrices."""

    def __init__(self, verticalFOV=45.0, nearClip=0.1, farClip=100.0):
        self.m_VerticalFOV, self.m_NearClip, self.m_FarClip = float(verticalFOV), float(nearClip), float(farClip)
        self.m_Position = np.array([0.0, 1.0, 2.0], np.float32)
        self.m_ForwardDirection = np.zeros(3, np.float32)
        lib().rt_host_camera_default_forward(ptr(self.m_ForwardDirection))
        self.m_ViewportWidth = self.m_ViewportHeight = 0
        self.m_Projection = np.eye(4, dtype=np.float32).reshape(16)
        self.m_InverseProjection = self.m_Projection.copy()
        self.m_View = self.m_Projection.copy()
        self.m_InverseView = self.m_Projection.copy()

    def _recalculate(self):
        w, h = max(self.m_ViewportWidth, 1), max(self.m_ViewportHeight, 1)
        lib().rt_host_camera_matrices(self.m_VerticalFOV, self.m_NearClip, self.m_FarClip, w, h, ptr(self.m_Position),
                                      ptr(self.m_ForwardDirection), ptr(self.m_Projection), ptr(self.m_InverseProjection),
                                      ptr(self.m_View), ptr(self.m_InverseView))

    def OnResize(self, width, height):
        if width == self.m_ViewportWidth and height == self.m_ViewportHeight:
            return
        self.m_ViewportWidth, self.m_ViewportHeight = int(width), int(height)
        self._recalculate()

    def SetPose(self, position, forward):
        """What Camera::OnUpdate does on input (Camera.cpp:96-100): new pose, matrices recomputed."""
        self.m_Position = np.ascontiguousarray(position, np.float32).copy()
        self.m_ForwardDirection = np.ascontiguousarray(forward, np.float32).copy()
        self._recalculate()

    def GetPosition(self): return self.m_Position
    def GetDirection(self): return self.m_ForwardDirection
    def GetProjection(self): return self.m_Projection
    def GetInverseProjection(self): return self.m_InverseProjection
    def GetView(self): return self.m_View
    def GetInverseView(self): return self.m_InverseView


class Image:
    """Stand-in for Walnut::Image (Walnut/Image.h:16-52) on the hand-off side: holds what SetData received."""

    def __init__(self, width, height):
        self._w, self._h = width, height
        self.data = np.zeros((height, width, 4), np.float32)

    def GetWidth(self): return self._w
    def GetHeight(self): return self._h

    def SetData(self, data):
        self.data = np.array(data, dtype=np.float32, copy=True)      # Walnut::Image::SetData copies out of the caller's buffer


class Renderer:
    """Renderer (Renderer.h:101-161) running on one MI355X."""

    def __init__(self, device=0, _borrowed=None):
        self._L = lib()
        self._owned = _borrowed is None
        self._h = self._L.rt_create(int(device)) if self._owned else _borrowed          # a group's context belongs to the group
        if not self._h:
            msg = self._L.rt_last_error(None)
            raise _capi.RtError("rt_create(%d) failed: %s" % (device, msg.decode() if msg else ""))
        self.device = int(device)
        self.m_settings = Settings()
        self.m_Image = None
        self.m_frameindex = 1
        self._mesh_ref = None           # the tree object whose mesh is resident (held, so its identity stays unique)
        self.download = True            # Render() ends in Image::SetData (host copy) unless switched off
        self.pipelined = False          # hand-off one frame late, the copy running beside the next trace (rt_map_image_pipelined)

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_owned", False):
            self._L.rt_destroy(self._h)
        self._h = None

    # ---- reference interface --------------------------------------------------------------------
    def GetSettings(self): return self.m_settings
    def GetImage(self): return self.m_Image
    def ResetFrameIndex(self): self.m_frameindex = 1
    def GetFrameIndex(self): return self.m_frameindex

    def OnResize(self, width, height):
        if self.m_Image is not None and self.m_Image.GetWidth() == width and self.m_Image.GetHeight() == height:
            return
        check(self._h, self._L.rt_resize(self._h, width, height), "rt_resize")
        self.m_Image = Image(width, height)
        self.ResetFrameIndex()

    def Render(self, scene, camera, frames=1):
        """One Renderer::Render (Renderer.cpp:45-135); `frames` > 1 runs that many consecutive Renders in one launch."""
        self.upload(scene, camera)
        check(self._h, self._L.rt_render_frames(self._h, self.m_frameindex, frames), "rt_render")
        if self.m_settings.Accumulate:
            self.m_frameindex += frames
        else:
            self.m_frameindex = 1
        if self.download:
            self.m_Image.SetData(self.map_image(self.pipelined))     # Image.SetData copies, as Walnut::Image::SetData does

    # ---- pieces ----------------------------------------------------------------------------------
    def upload(self, scene, camera):
        L, h = self._L, self._h
        st = self.m_settings.to_c()
        check(h, L.rt_set_settings(h, C.byref(st)), "rt_set_settings")
        mats = np.ascontiguousarray(scene.materials, MATERIAL_DTYPE)
        check(h, L.rt_set_materials(h, ptr(mats), len(mats)), "rt_set_materials")
        sph = np.ascontiguousarray(scene.spheres, SPHERE_DTYPE)
        check(h, L.rt_set_spheres(h, ptr(sph), len(sph)), "rt_set_spheres")
        if scene.kd_tree is not None and self._mesh_ref is not scene.kd_tree:       # mesh + tree go up once
            check(h, L.rt_set_mesh(h, ptr(scene.verts), len(scene.verts), ptr(scene.tri_idx), ptr(scene.normals9),
                                   ptr(scene.tri_mat), len(scene.tri_idx)), "rt_set_mesh")
            check(h, L.rt_set_kdtree(h, C.byref(scene.kd_tree.flat)), "rt_set_kdtree")
            self._mesh_ref = scene.kd_tree
        check(h, L.rt_set_camera(h, ptr(camera.GetPosition()), ptr(camera.GetInverseProjection()),
                                 ptr(camera.GetInverseView())), "rt_set_camera")

    def render_only(self, frame_index, frames=1):
        check(self._h, self._L.rt_render_frames(self._h, frame_index, frames), "rt_render")

    def resolve(self, frame_index):
        check(self._h, self._L.rt_resolve(self._h, frame_index), "rt_resolve")

    def reset_accum(self):
        check(self._h, self._L.rt_reset_accum(self._h), "rt_reset_accum")

    def sync(self):
        check(self._h, self._L.rt_sync(self._h), "rt_sync")

    def set_bands(self, index, count):
        check(self._h, self._L.rt_set_bands(self._h, index, count), "rt_set_bands")

    def set_stream(self, hip_stream):
        check(self._h, self._L.rt_set_stream(self._h, hip_stream), "rt_set_stream")

    def set_ray_dirs(self, dirs):
        if dirs is None:
            check(self._h, self._L.rt_set_ray_dirs(self._h, None, 0), "rt_set_ray_dirs")
            return
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        check(self._h, self._L.rt_set_ray_dirs(self._h, ptr(d), len(d)), "rt_set_ray_dirs")

    def record_ids(self, on=True):
        check(self._h, self._L.rt_set_record_ids(self._h, int(on)), "rt_set_record_ids")

    def push_settings(self):
        """Send GetSettings() to the library without the rest of upload()."""
        st = self.m_settings.to_c()
        check(self._h, self._L.rt_set_settings(self._h, C.byref(st)), "rt_set_settings")

    def traversal_in_use(self):
        """_capi.ORDER_REFERENCE or ORDER_FRONT_TO_BACK: what the current settings and tree resolve to (after upload)."""
        self.push_settings()
        return self._L.rt_traversal_in_use(self._h)

    def set_stats(self, on=True):
        check(self._h, self._L.rt_set_stats(self._h, int(on)), "rt_set_stats")

    def _size(self):
        return self.m_Image.GetWidth(), self.m_Image.GetHeight()

    def read_image(self):
        w, h = self._size()
        out = np.empty((h, w, 4), np.float32)
        check(self._h, self._L.rt_read_image(self._h, ptr(out)), "rt_read_image")
        return out

    def map_image(self, pipelined=False):
        """The frame in the library's page-locked staging buffer, as an (h, w, 4) view (valid until the next map_image /
        OnResize): what a display path hands to Image::SetData without another copy.  `pipelined`: the frame of the
        previous call, while this one's copy runs beside the next trace."""
        w, h = self._size()
        p = (self._L.rt_map_image_pipelined if pipelined else self._L.rt_map_image)(self._h)
        if not p:
            check(self._h, -1, "rt_map_image")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(h, w, 4))

    def read_accum(self):
        w, h = self._size()
        out = np.empty((h, w, 3), np.float32)
        check(self._h, self._L.rt_read_accum(self._h, ptr(out)), "rt_read_accum")
        return out

    def read_primary_ids(self):
        w, h = self._size()
        out = np.empty((h, w), np.int32)
        check(self._h, self._L.rt_read_primary_ids(self._h, ptr(out)), "rt_read_primary_ids")
        return out

    def read_ray_dirs(self):
        w, h = self._size()
        out = np.empty((h * w, 3), np.float32)
        check(self._h, self._L.rt_read_ray_dirs(self._h, ptr(out)), "rt_read_ray_dirs")
        return out

    def read_counters(self):
        c = RtCounters()
        check(self._h, self._L.rt_read_counters(self._h, C.byref(c)), "rt_read_counters")
        return c

    def read_wave_clock(self, max_waves=16384):
        """(n, 8) uint64 per wave of the last trace launch made with set_stats(2 or 3): start, end (100 MHz ticks), main-loop
        iterations, walk-loop trips, sum over trips of the walking lanes (set_stats(3) only), ticks inside the walk loop, ticks
        spent waiting for paths of other waves, ticks in the shading block."""
        out = np.zeros((max_waves, 8), np.uint64)
        n = C.c_uint32()
        check(self._h, self._L.rt_read_wave_clock(self._h, ptr(out), max_waves, C.byref(n)), "rt_read_wave_clock")
        return out[:n.value]

    def reset_counters(self):
        check(self._h, self._L.rt_reset_counters(self._h), "rt_reset_counters")

    def kernel_time(self, reset=False):
        t = RtKernelTime()
        check(self._h, self._L.rt_get_kernel_time(self._h, C.byref(t)), "rt_get_kernel_time")
        if reset:
            check(self._h, self._L.rt_reset_kernel_time(self._h), "rt_reset_kernel_time")
        return t

    def set_kernel_timing(self, every):
        """Events around every `every`-th launch only (0: none); see rt_set_kernel_timing."""
        check(self._h, self._L.rt_set_kernel_timing(self._h, int(every)), "rt_set_kernel_timing")

    def device_accum_ptr(self):
        return self._L.rt_device_accum(self._h)

    def device_image_ptr(self):
        return self._L.rt_device_image(self._h)


class GroupRenderer:
    """Renderer (Renderer.h:101-161) over several GPUs of one node from ONE process (rt_group_*): same members as Renderer.
    split = "tiles": every Render is one frame, traced as 16-row bands dealt round-robin over the devices (single-sample
    latency); split = "samples": every Render is len(devices) * frames_per_device frames, dealt by frame index (progressive
    accumulation).  Device 0 of the list assembles the frame.  A device may be listed more than once."""

    def __init__(self, devices, split="tiles", frames_per_device=1):
        assert split in ("tiles", "samples")
        self._L = lib()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._g = self._L.rt_group_create(ids, len(devices))
        if not self._g:
            msg = self._L.rt_last_error(None)
            raise _capi.RtError("rt_group_create(%s) failed: %s" % (list(devices), msg.decode() if msg else ""))
        self.split, self.frames_per_device = split, int(frames_per_device)
        self.members = [Renderer(d, _borrowed=self._L.rt_group_ctx(self._g, i)) for i, d in enumerate(devices)]
        self.m_settings = Settings()
        self.m_Image = None
        self.m_frameindex = 1
        self.download = True

    def __del__(self):
        if getattr(self, "_g", None):
            self.members = []
            self._L.rt_group_destroy(self._g)
            self._g = None

    def GetSettings(self): return self.m_settings
    def GetImage(self): return self.m_Image
    def ResetFrameIndex(self): self.m_frameindex = 1
    def GetFrameIndex(self): return self.m_frameindex

    def OnResize(self, width, height):
        if self.m_Image is not None and self.m_Image.GetWidth() == width and self.m_Image.GetHeight() == height:
            return
        for m in self.members:
            m.OnResize(width, height)
        self.m_Image = Image(width, height)
        self.ResetFrameIndex()

    def _gcheck(self, rc, what):
        if rc != _capi.RT_OK:
            msg = self._L.rt_group_last_error(self._g)
            raise _capi.RtError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def Render(self, scene, camera):
        for m in self.members:
            m.m_settings = self.m_settings
            m.upload(scene, camera)
        if self.split == "tiles":
            frames = 1
            self._gcheck(self._L.rt_group_render_tiles(self._g, self.m_frameindex, 1), "rt_group_render_tiles")
        else:
            frames = len(self.members) * self.frames_per_device
            self._gcheck(self._L.rt_group_render_samples(self._g, self.m_frameindex, self.frames_per_device), "rt_group_render_samples")
        self.m_frameindex = self.m_frameindex + frames if self.m_settings.Accumulate else 1
        if self.download:
            self.m_Image.SetData(self.members[0].map_image())

    def read_accum(self):
        """The accumulated radiance of the whole frame: gathered from the owners of the bands (tiles), or the sum over the
        devices (samples)."""
        from .dist import band_rows
        w, h = self.m_Image.GetWidth(), self.m_Image.GetHeight()
        if self.split == "samples":
            out = np.empty((h, w, 3), np.float32)
            self._gcheck(self._L.rt_group_read_accum(self._g, ptr(out)), "rt_group_read_accum")
            return out
        out = np.zeros((h, w, 3), np.float32)
        for i, m in enumerate(self.members):
            rows = band_rows(h, i, len(self.members))
            out[rows] = m.read_accum()[rows]
        return out
