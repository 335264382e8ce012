"""rt_render_spheres_noalloc (the entry point whose working memory is a static device array of the library) against rt_render,
through the C ABI: accumulation and frame bit-exact, on two calls in a row, the second with a smaller input -- a smaller
frame, fewer spheres and materials, another camera -- so that whatever the first call left in the array must not matter.
The caller's buffers are torch tensors here; the entry point itself allocates nothing."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _reference(rt, scene, cam, w, h, frames, bounces, aa):
    r = rt.Renderer(0)
    st = r.GetSettings()
    st.SceneMode, st.Bounces, st.AntiAliasing, st.UseACE_Color, st.Accumulate, st.Seed = 1, bounces, aa, True, True, 1337
    r.OnResize(w, h)
    for _ in range(frames):
        r.Render(scene, cam)
    return r.read_accum().reshape(-1, 3), r.GetImage().data.reshape(-1, 4).copy()


def _noalloc(rt, torch, scene, cam, w, h, frames, bounces, aa, stream):
    from raytracer_amd import _capi
    from raytracer_amd.scenes import MATERIAL_DTYPE, SPHERE_DTYPE
    L = _capi.lib()
    acc = torch.full((h, w, 4), 7.0, dtype=torch.float32, device="cuda:0")       # frame index 1 must not read it
    img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    sph = np.ascontiguousarray(scene.spheres, SPHERE_DTYPE); mats = np.ascontiguousarray(scene.materials, MATERIAL_DTYPE)
    st = _capi.RtSettings(1, 1, 1, int(aa), bounces, 1337, 0)
    torch.cuda.synchronize()
    for f in range(1, frames + 1):
        rc = L.rt_render_spheres_noalloc(0, stream, _capi.ptr(sph), len(sph), _capi.ptr(mats), len(mats), C.byref(st),
                                         _capi.ptr(cam.GetPosition()), _capi.ptr(cam.GetInverseProjection()), _capi.ptr(cam.GetInverseView()),
                                         w, h, f, C.c_void_p(acc.data_ptr()), C.c_void_p(img.data_ptr()))
        assert rc == 0, L.rt_last_error(None)
    torch.cuda.synchronize()
    return acc.cpu().numpy().reshape(-1, 4)[:, :3], img.cpu().numpy().reshape(-1, 4)


def test_noalloc_entry_point_matches_rt_render_twice_in_a_row():
    import torch
    import raytracer_amd as rt
    from raytracer_amd import _capi, scenes
    L = _capi.lib()
    stream = torch.cuda.Stream(device="cuda:0")
    handle = C.c_void_p(stream.cuda_stream)

    big = rt.Scene(); big.spheres = scenes.default_spheres(); big.materials = scenes.default_materials()
    cam1 = rt.Camera(70.0, 0.05, 100.0); cam1.OnResize(320, 200)
    small = rt.Scene(); small.spheres = scenes.default_spheres()[[0, 1, 2, 3, 5, 8]].copy()       # three walls less, the light stays
    small.materials = scenes.default_materials()[:6].copy()
    assert small.spheres["material"].max() < 6
    cam2 = rt.Camera(55.0, 0.05, 100.0); cam2.OnResize(96, 64)
    cam2.SetPose(np.array([0.3, 1.2, 2.5], np.float32), np.array([-0.1, -0.2, -1.0], np.float32) / np.float32(np.sqrt(1.05)))

    with torch.cuda.stream(stream):
        for scene, cam, w, h, frames, bounces, aa in ((big, cam1, 320, 200, 3, 5, False), (small, cam2, 96, 64, 2, 4, True)):
            acc, img = _noalloc(rt, torch, scene, cam, w, h, frames, bounces, aa, handle)
            want_acc, want_img = _reference(rt, scene, cam, w, h, frames, bounces, aa)
            assert np.array_equal(bits(acc), bits(want_acc)), "accumulation differs at %dx%d" % (w, h)
            assert np.all(np.abs(acc - want_acc) <= 1e-5 + 1e-5 * np.abs(want_acc))         # the stated tolerance, trivially met
            assert np.array_equal(bits(img), bits(want_img)), "frame differs at %dx%d" % (w, h)
            assert (want_acc.sum(axis=1) > 0).mean() > 0.02         # light reaches the frame

    # beyond the fixed limits: a clear error, nothing launched
    st = _capi.RtSettings(1, 1, 1, 0, 5, 1337, 0)
    sph = np.ascontiguousarray(big.spheres); mats = np.ascontiguousarray(big.materials)
    buf = torch.zeros(16, device="cuda:0")
    args = (_capi.ptr(sph), len(sph), _capi.ptr(mats), len(mats), C.byref(st), _capi.ptr(cam1.GetPosition()),
            _capi.ptr(cam1.GetInverseProjection()), _capi.ptr(cam1.GetInverseView()))
    rc = L.rt_render_spheres_noalloc(0, handle, *args, 4096, 4096, 1, C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()))
    assert rc == -5 and b"exceed" in L.rt_last_error(None)
    many = np.concatenate([sph] * 4)
    rc = L.rt_render_spheres_noalloc(0, handle, _capi.ptr(many), len(many), *args[2:], 64, 64, 1, C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()))
    assert rc == -5


def test_noalloc_allocates_nothing_replays_from_a_graph_and_equals_the_oracle(orc):
    """What the entry point exists for: after its set-up on the device, calls change neither the device's free memory nor need
    anything a stream capture forbids -- three Renders captured into a HIP graph (torch's capture runs in the mode that fails any
    allocation made meanwhile) replay to the same bits.  Those bits are the oracle's.  A scene without spheres is a frame of misses."""
    import torch
    import raytracer_amd as rt
    from raytracer_amd import _capi, scenes
    from raytracer_amd.scenes import MATERIAL_DTYPE, SPHERE_DTYPE
    L = _capi.lib()
    assert L.rt_render_spheres_noalloc_setup(0) == 0, L.rt_last_error(None)
    w, h, frames, bounces = 160, 96, 3, 5
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(w, h)
    sph = np.ascontiguousarray(scenes.default_spheres(), SPHERE_DTYPE); mats = np.ascontiguousarray(scenes.default_materials(), MATERIAL_DTYPE)
    st = _capi.RtSettings(1, 1, 1, 0, bounces, 1337, 0, 0)
    acc = torch.full((h, w, 4), 7.0, dtype=torch.float32, device="cuda:0")
    img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")

    def calls(stream_handle, spheres=sph):
        for f in range(1, frames + 1):
            rc = L.rt_render_spheres_noalloc(0, stream_handle, _capi.ptr(spheres) if len(spheres) else None, len(spheres), _capi.ptr(mats), len(mats),
                                             C.byref(st), _capi.ptr(cam.GetPosition()), _capi.ptr(cam.GetInverseProjection()),
                                             _capi.ptr(cam.GetInverseView()), w, h, f, C.c_void_p(acc.data_ptr()), C.c_void_p(img.data_ptr()))
            assert rc == 0, L.rt_last_error(None)

    # the oracle's three frames
    osc = orc.Scene(); osc.set_spheres(sph); osc.set_materials(mats)
    dirs = orc.ray_dirs(cam.GetInverseProjection(), cam.GetInverseView(), w, h)
    want_acc = np.zeros((h * w, 3), np.float32); want_img = np.zeros((h * w, 4), np.float32)
    for f in range(1, frames + 1):
        osc.render_frame(orc.Settings(1, 1, 1, 0, bounces), cam.GetPosition(), dirs, w, h, f, want_acc, want_img)

    def check(what):
        torch.cuda.synchronize()
        assert np.array_equal(bits(acc.cpu().numpy().reshape(-1, 4)[:, :3]), bits(want_acc)), what
        assert np.array_equal(bits(img.cpu().numpy().reshape(-1, 4)), bits(want_img)), what

    stream = torch.cuda.Stream(device="cuda:0")
    with torch.cuda.stream(stream):
        calls(C.c_void_p(stream.cuda_stream))              # (an ordinary call first: whatever the runtime sets up lazily is set up)
        check("first calls")
        free_before = torch.cuda.mem_get_info(0)[0]
        calls(C.c_void_p(stream.cuda_stream))
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free_before, "the calls changed the device's free memory"
        check("calls after the set-up")
    # captured: nothing but kernel launches may happen in there
    graph = torch.cuda.CUDAGraph()
    acc.fill_(3.0); img.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        calls(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for replay in range(2):
        acc.fill_(5.0); img.zero_()
        graph.replay()
        check("graph replay %d" % replay)
    # no spheres at all: every ray misses
    none = np.zeros(0, SPHERE_DTYPE)
    with torch.cuda.stream(stream):
        calls(C.c_void_p(stream.cuda_stream), none)
    torch.cuda.synchronize()
    osc.set_spheres(none)
    want_acc[:] = 0; want_img[:] = 0
    for f in range(1, frames + 1):
        osc.render_frame(orc.Settings(1, 1, 1, 0, bounces), cam.GetPosition(), dirs, w, h, f, want_acc, want_img)
    check("a scene without spheres")
    assert not want_acc.any()
