"""Splitting the frame over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Two partitions, both with replicated read-only scene data and no exchange between frames:
  * sample split (progressive accumulation): rank r renders frame indices r+1, r+1+N, ... into its own accumulation
    buffer; one sum of the buffers at the end (all-reduce), then the image is resolved from the summed buffer.
  * tile split (single frames): rank r renders the 16-row bands b with b % N == r; one gather of the bands per
    displayed frame.
The functions take torch tensors, so the same code runs on CPU tensors under the gloo backend (tests) and on device
tensors that alias the renderer's buffers under RCCL.
"""
import numpy as np

BAND_ROWS = 16          # RT_TILE: the height of one workgroup's pixel tile


def frames_for_rank(rank, world, steps, first=1):
    """Frame indices (1-based) rank `rank` renders in a sample split of `steps` frames per rank."""
    return [first + rank + i * world for i in range(steps)]


def band_rows(height, rank, world):
    """Image rows owned by `rank` in a tile split."""
    rows = np.arange(height)
    return rows[(rows // BAND_ROWS) % world == rank]


class DeviceArray:
    """Raw device memory presented through __cuda_array_interface__ so torch can alias it without a copy."""

    def __init__(self, dev_ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(dev_ptr), False),
                                         "version": 2, "strides": None}


def accum_tensor(renderer):
    """The renderer's accumulation buffer as a (h, w, 4) float32 device tensor (aliases, no copy)."""
    import torch
    w, h = renderer.m_Image.GetWidth(), renderer.m_Image.GetHeight()
    return torch.as_tensor(DeviceArray(renderer.device_accum_ptr(), (h, w, 4)), device="cuda:%d" % renderer.device)


def image_tensor(renderer):
    import torch
    w, h = renderer.m_Image.GetWidth(), renderer.m_Image.GetHeight()
    return torch.as_tensor(DeviceArray(renderer.device_image_ptr(), (h, w, 4)), device="cuda:%d" % renderer.device)


def sum_accumulation(accum, group=None):
    """Sample split: sum the per-rank accumulation buffers in place on every rank."""
    import torch.distributed as dist
    if dist.is_initialized():
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum


def gather_bands(image, rank, world, dst=0, group=None):
    """Tile split: collect every rank's bands of `image` ((h, w, c) tensor, only this rank's rows valid) on `dst`.
    Returns the assembled image on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    h = image.shape[0]
    if world == 1:
        return image
    counts = [len(band_rows(h, r, world)) for r in range(world)]
    most = max(counts)
    mine = torch.as_tensor(band_rows(h, rank, world), device=image.device, dtype=torch.long)
    send = torch.zeros((most,) + tuple(image.shape[1:]), dtype=image.dtype, device=image.device)
    send[: len(mine)] = image.index_select(0, mine)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty_like(image)
    for r in range(world):
        rows = torch.as_tensor(band_rows(h, r, world), device=image.device, dtype=torch.long)
        out.index_copy_(0, rows, recv[r][: counts[r]])
    return out
