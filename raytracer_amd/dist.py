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


def sum_accumulation(accum, dst=0, group=None):
    """Sample split: the per-rank accumulation buffers summed onto rank `dst` (the only rank that resolves the image).
    `accum` is (h, w, 4) with an unused fourth component or (n, 3): only the three sums travel (99.5 MB at 4K, not 133).
    `dst` is a GLOBAL rank, as torch.distributed.reduce takes it, also when `group` is a sub-group.  Only `dst` holds the total
    afterwards: the other ranks' buffers keep their own sums."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return accum
    if accum.shape[-1] == 4:
        rgb = accum[..., :3].contiguous()
        dist.reduce(rgb, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank() == dst:                          # (the global rank: dst is one)
            accum[..., :3] = rgb
    else:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


def dist_ready():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def dist_rank():
    import torch.distributed as dist
    return dist.get_rank()


class BandGather:
    """Tile split: collects every rank's 16-row bands of `image` ((h, w, c) tensor, only this rank's rows valid) on `dst`,
    once per displayed frame.  Row indices and the send / receive buffers are set up once.  `rank` is this process's place among
    the `world` band owners (its rank within `group`); `dst` is a GLOBAL rank, as torch.distributed.gather takes it.
    `image` may be a callable returning the tensor: a renderer's frame buffer moves on rt_resize and at every
    rt_map_image_pipelined, and a tensor that aliased it when this object was built would then be stale."""

    def __init__(self, image, rank, world, dst=0, group=None):
        import torch
        self._image = image
        image = self.image
        self.rank, self.world, self.dst, self.group = rank, world, dst, group
        h = image.shape[0]
        self.counts = [len(band_rows(h, r, world)) for r in range(world)]
        most = max(self.counts)
        dev = image.device
        self.mine = torch.as_tensor(band_rows(h, rank, world), device=dev, dtype=torch.long)
        self.send = torch.zeros((most,) + tuple(image.shape[1:]), dtype=image.dtype, device=dev)
        self.recv = self.rows = None
        is_dst = (dist_rank() == dst) if dist_ready() else rank == dst
        if is_dst and world > 1:
            self.recv = [torch.empty_like(self.send) for _ in range(world)]
            self.rows = [torch.as_tensor(band_rows(h, r, world), device=dev, dtype=torch.long) for r in range(world)]

    @property
    def image(self):
        return self._image() if callable(self._image) else self._image

    def gather(self):
        """Returns the assembled image on dst (in place: the other ranks' rows are written into `image`), None elsewhere."""
        import torch
        import torch.distributed as dist
        image = self.image
        if self.world == 1:
            return image
        torch.index_select(image, 0, self.mine, out=self.send[: len(self.mine)])
        dist.gather(self.send, self.recv if self.recv is not None else None, dst=self.dst, group=self.group)
        if self.recv is None:
            return None
        for r in range(self.world):
            if r != self.rank:
                image.index_copy_(0, self.rows[r], self.recv[r][: self.counts[r]])
        return image


def gather_bands(image, rank, world, dst=0, group=None):
    """One-off form of BandGather (sets everything up for a single gather)."""
    return BandGather(image, rank, world, dst, group).gather()
