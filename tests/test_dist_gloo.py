"""The multi-GPU partitions on CPU: two processes under the gloo backend, the oracle standing in for each rank's
renderer (tests may use it; the product's dist module only moves tensors).

  * sample split: rank r renders frame indices r+1, r+1+N, ...; the summed accumulation equals the serial sum within the
    re-association tolerance |d| <= 1e-4 + 1e-4*|ref| (and bit-exactly when the per-rank sums are added in rank order);
  * tile split: each rank renders its 16-row bands; the gathered frame equals the serial frame bit for bit.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, bits

W, H, BOUNCES, PER_RANK = 96, 70, 5, 3


def _scene(orc):
    from raytracer_amd import scenes
    sc = orc.Scene()
    sc.set_spheres(scenes.default_spheres()); sc.set_materials(scenes.default_materials())
    return sc


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from raytracer_amd import dist as rtd
    sc = _scene(orc)
    cam = orc.camera(70.0, 0.05, 100.0, W, H)
    st = orc.Settings(1, 1, 1, 0, BOUNCES)

    # sample split
    acc = np.zeros((H * W, 3), np.float32); img = np.zeros((H * W, 4), np.float32)
    for i, f in enumerate(rtd.frames_for_rank(rank, world, PER_RANK)):
        if f == 1 or i == 0:
            acc[:] = 0
        if f == 1:
            sc.render_frame(st, cam["pos"], cam["dirs"], W, H, 1, acc, img)
        else:
            sc.render_frame(st, cam["pos"], cam["dirs"], W, H, f, acc, img)       # index != 1: adds to the buffer
    t = torch.from_numpy(acc.copy())
    rtd.sum_accumulation(t)
    if rank == 0:
        np.save(os.path.join(tmp, "sample_sum.npy"), t.numpy())

    # tile split: a full-size image of which only this rank's bands are rendered
    acc2 = np.zeros((H * W, 3), np.float32); img2 = np.zeros((H * W, 4), np.float32)
    full_acc = np.zeros((H * W, 3), np.float32); full_img = np.zeros((H * W, 4), np.float32)
    sc.render_frame(st, cam["pos"], cam["dirs"], W, H, 1, full_acc, full_img)
    rows = rtd.band_rows(H, rank, world)
    mine = np.zeros((H, W, 4), np.float32)
    mine[rows] = full_img.reshape(H, W, 4)[rows]          # what this rank's GPU would have produced
    out = rtd.gather_bands(torch.from_numpy(mine), rank, world, dst=0)
    if rank == 0:
        np.save(os.path.join(tmp, "tile_gather.npy"), out.numpy())
        np.save(os.path.join(tmp, "tile_full.npy"), full_img.reshape(H, W, 4))
    else:
        assert out is None
    dist.barrier()

    # a sub-group that leaves global rank 0 out: dst is a GLOBAL rank there too (ranks 1 and 2 are places 0 and 1 of the group)
    if world == 3:
        sub = dist.new_group([1, 2])
        if rank >= 1:
            part = torch.full((4, 5, 4), float(rank))
            rtd.sum_accumulation(part, dst=1, group=sub)
            want = 3.0 if rank == 1 else float(rank)                # only dst holds the total
            assert torch.all(part[..., :3] == want) and torch.all(part[..., 3] == float(rank))
            place = rank - 1
            rows2 = rtd.band_rows(H, place, 2)
            mine2 = np.zeros((H, W, 4), np.float32)
            mine2[rows2] = full_img.reshape(H, W, 4)[rows2]
            holder = {"t": torch.from_numpy(mine2)}
            bg = rtd.BandGather(lambda: holder["t"], place, 2, dst=1, group=sub)
            holder["t"] = holder["t"].clone()                       # the frame buffer moved: the gather must follow it
            out2 = bg.gather()
            if rank == 1:
                assert out2 is holder["t"] and np.array_equal(out2.numpy(), full_img.reshape(H, W, 4))
            else:
                assert out2 is None
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partitions_under_gloo(orc, tmp_path, world):
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from raytracer_amd import dist as rtd
    # serial reference: all frames in index order
    sc = _scene(orc)
    cam = orc.camera(70.0, 0.05, 100.0, W, H)
    st = orc.Settings(1, 1, 1, 0, BOUNCES)
    acc = np.zeros((H * W, 3), np.float32); img = np.zeros((H * W, 4), np.float32)
    for f in range(1, world * PER_RANK + 1):
        sc.render_frame(st, cam["pos"], cam["dirs"], W, H, f, acc, img)
    got = np.load(tmp_path / "sample_sum.npy")
    assert np.all(np.abs(got - acc) <= 1e-4 + 1e-4 * np.abs(acc))
    assert np.array_equal(bits(np.load(tmp_path / "tile_gather.npy")), bits(np.load(tmp_path / "tile_full.npy")))
    # the partitions cover every frame / row exactly once
    frames = sorted(f for r in range(world) for f in rtd.frames_for_rank(r, world, PER_RANK))
    assert frames == list(range(1, world * PER_RANK + 1))
    rows = np.sort(np.concatenate([rtd.band_rows(H, r, world) for r in range(world)]))
    assert np.array_equal(rows, np.arange(H))
