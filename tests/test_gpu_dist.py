"""The multi-process path on the product: two processes, one process group (gloo), each rank running the HIP renderer on the GPU
for its own part, the exchange through raytracer_amd.dist as bench.py --gpus N makes it (there over RCCL on device tensors; two
ranks cannot share one GPU under RCCL, so on a one-GPU box the tensors go through host memory).

  * sample split: rank r renders frame indices r+1, r+1+N, ... into its own buffer, the sums are reduced onto rank 0 and resolved
    there: against the oracle's frame-by-frame sum within the re-association tolerance |d| <= 1e-4 + 1e-4 |ref|;
  * tile split with the 3x3 filter on: each rank renders its 16-row bands (and the rows bordering them), the bands of the frame
    are gathered on rank 0: bit-exact against the oracle's filtered frame.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, bits

pytestmark = pytest.mark.gpu
W, H, BOUNCES, PER_RANK, WORLD = 112, 80, 5, 3, 2


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracer_amd as rt
    from raytracer_amd import dist as rtd, scenes
    scene = rt.Scene(); scene.spheres = scenes.default_spheres()
    verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(24, mat_offset=7)
    scene.materials = np.concatenate([scenes.default_materials(), mesh_mats])
    scene.set_triangles(verts, tris, normals9, tri_mat, nthreads=2)
    cam = rt.Camera(70.0, 0.05, 100.0); cam.OnResize(W, H)

    # sample split (spheres + mesh)
    r = rt.Renderer(0); r.download = False
    st = r.GetSettings(); st.SceneMode, st.Bounces = 2, BOUNCES
    r.OnResize(W, H); r.upload(scene, cam); r.reset_accum()
    for f in rtd.frames_for_rank(rank, world, PER_RANK):
        r.render_only(f)
    acc = torch.from_numpy(r.read_accum().copy())
    rtd.sum_accumulation(acc, dst=0)
    if rank == 0:
        np.save(os.path.join(tmp, "sample_sum.npy"), acc.numpy())

    # tile split with the filter
    r2 = rt.Renderer(0)
    st = r2.GetSettings(); st.SceneMode, st.Bounces, st.AntiAliasing = 2, BOUNCES, True
    r2.OnResize(W, H); r2.set_bands(rank, world)
    for _ in range(2):
        r2.Render(scene, cam)
    img = torch.from_numpy(r2.GetImage().data.copy())
    out = rtd.BandGather(img, rank, world).gather()
    if rank == 0:
        np.save(os.path.join(tmp, "tile_image.npy"), out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_render_their_parts_on_the_gpu(orc, tmp_path):
    from raytracer_amd import scenes
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    verts, tris, normals9, tri_mat, mesh_mats = scenes.displaced_grid(24, mat_offset=7)
    osc = orc.Scene()
    osc.set_spheres(scenes.default_spheres()); osc.set_materials(np.concatenate([scenes.default_materials(), mesh_mats]))
    osc.set_mesh(verts, tris, normals9, tri_mat); osc.build_kd()
    cam = orc.camera(70.0, 0.05, 100.0, W, H)
    acc = np.zeros((H * W, 3), np.float32); img = np.zeros((H * W, 4), np.float32)
    for f in range(1, WORLD * PER_RANK + 1):
        osc.render_frame(orc.Settings(1, 2, 1, 0, BOUNCES), cam["pos"], cam["dirs"], W, H, f, acc, img)
    got = np.load(tmp_path / "sample_sum.npy").reshape(-1, 3)
    assert np.all(np.abs(got - acc) <= 1e-4 + 1e-4 * np.abs(acc))
    assert (acc.sum(axis=1) > 0).mean() > 0.05
    acc2 = np.zeros((H * W, 3), np.float32); img2 = np.zeros((H * W, 4), np.float32)
    for f in (1, 2):
        osc.render_frame(orc.Settings(1, 2, 1, 1, BOUNCES), cam["pos"], cam["dirs"], W, H, f, acc2, img2)
    assert np.array_equal(bits(np.load(tmp_path / "tile_image.npy").reshape(-1, 4)), bits(img2))
