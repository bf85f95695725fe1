"""Multi-GPU host side: one process per GPU, 8x8-pixel tiles dealt round-robin, and the ONE
exchange step of the path — the final framebuffer gather to rank 0 (RCCL over xGMI when the
tensors live on MI355X, gloo on CPU in tests).  No all-reduce: every pixel is produced by
exactly one rank, so the collective is a gather of equal-sized tile slabs.

DeviceFramebufferGather is the product path: slabs are packed and unpacked by the library's own
tile kernels (vk_pack_tiles_device / vk_unpack_tiles_device), so a step is render -> pack ->
dist.gather -> unpack with no torch indexing op in it.  FramebufferGather is the index-based
twin for tensors that never were on a GPU (the CPU tests, where the oracle stands in for the
renderer).
"""
import numpy as np

TILE = 8


def tile_pixel_indices(width, height, rank, world, top_down=False):
    """Flat pixel indices (y*width + x, y = 0 bottom) of the tiles owned by `rank`, tile by tile
    in the order the kernel deals them (tile t -> rank t % world).  top_down: indices into an image stored
    top row first (VK_OUTPUT_RGB8, main.rs:209) instead."""
    tiles_x = (width + TILE - 1) // TILE
    tiles_y = (height + TILE - 1) // TILE
    out = []
    ys, xs = np.mgrid[0:TILE, 0:TILE]
    for t in range(rank, tiles_x * tiles_y, world):
        tx, ty = (t % tiles_x) * TILE, (t // tiles_x) * TILE
        x, y = (tx + xs).ravel(), (ty + ys).ravel()
        ok = (x < width) & (y < height)
        row = (height - 1 - y[ok]) if top_down else y[ok]
        out.append((row * width + x[ok]).astype(np.int64))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


class FramebufferGather:
    """Pre-computes the per-rank index sets once; `gather(fb)` moves each rank's tile slab to
    rank 0 with a single torch.distributed.gather and scatters it into the full image."""

    def __init__(self, width, height, rank, world, device, stage_on_cpu=False, rgb8=False):
        """stage_on_cpu: exchange through host tensors (gloo rehearsals: gloo cannot gather GPU tensors).
        rgb8: the framebuffers are VK_OUTPUT_RGB8 images (uint8, top row first): the slabs are a quarter the size
        (SURVEY §8f-1)."""
        import torch
        self.torch = torch
        self.stage_on_cpu = stage_on_cpu
        self.width, self.height, self.rank, self.world = width, height, rank, world
        dtype = torch.uint8 if rgb8 else torch.float32
        idx = [tile_pixel_indices(width, height, r, world, top_down=rgb8) for r in range(world)]
        self.slab_len = max(len(i) for i in idx)
        self.own = torch.from_numpy(idx[rank]).to(device)
        self.all_idx = [torch.from_numpy(i).to(device) for i in idx] if rank == 0 else None
        xdev = "cpu" if stage_on_cpu else device
        self.slab = torch.zeros((self.slab_len, 3), dtype=dtype, device=xdev)
        self.recv = [torch.zeros((self.slab_len, 3), dtype=dtype, device=xdev) for _ in range(world)] if rank == 0 else None

    def gather(self, fb, out=None):
        """fb: (height, width, 3) tensor (float32, or uint8 for rgb8) holding this rank's tiles.  Returns the full
        image on rank 0 (written into `out` if given), None elsewhere."""
        import torch.distributed as dist
        flat = fb.view(-1, 3)
        n = self.own.numel()
        self.slab[:n] = flat.index_select(0, self.own).to(self.slab.device)
        if self.world == 1:
            if out is None:
                return fb
            out.copy_(fb)
            return out
        dist.gather(self.slab, self.recv if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        full = out if out is not None else self.torch.zeros_like(fb)
        ff = full.view(-1, 3)
        for r in range(self.world):
            m = self.all_idx[r].numel()
            ff.index_copy_(0, self.all_idx[r], self.recv[r][:m].to(ff.device))
        return full


class DeviceFramebufferGather:
    """One step's exchange for a one-process-per-GPU host.  `scene` is this rank's DeviceScene; `fb` the f32 framebuffer its
    vk_render_device call wrote (this rank's tiles).  gather(fb, out) packs the rank's slab on the device (floats, or bytes through
    Vec3::to_color when rgb8), moves the slabs to rank 0 with ONE torch.distributed.gather (RCCL when the tensors are on the GPU;
    stage_on_cpu: through host tensors, for gloo rehearsals) and unpacks them there into `out`: (height, width, 3) float32 with
    y = 0 the bottom row, or uint8 with row 0 the top row (VK_OUTPUT_RGB8)."""

    def __init__(self, scene, width, height, rank, world, device, stage_on_cpu=False, rgb8=False):
        import torch
        from . import ffi
        self.torch, self.scene = torch, scene
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.fmt = ffi.VK_OUTPUT_RGB8 if rgb8 else ffi.VK_OUTPUT_F32
        self.stage_on_cpu = stage_on_cpu
        lib = ffi.load_device_lib()
        # the largest rank's; at least one byte: an image with fewer 8x8 tiles than ranks leaves some ranks (or all but one) without a
        # tile, and a zero-sized tensor has a null data pointer, which vk_pack_tiles_device rejects
        self.slab_bytes = max(1, int(lib.vk_tile_slab_bytes(width, height, self.fmt, world, world)))
        self.slab = torch.zeros(self.slab_bytes, dtype=torch.uint8, device=device)
        xdev = "cpu" if stage_on_cpu else device
        self.xslab = torch.zeros(self.slab_bytes, dtype=torch.uint8, device=xdev) if stage_on_cpu else self.slab
        self.recv = [torch.zeros(self.slab_bytes, dtype=torch.uint8, device=xdev) for _ in range(world)] if rank == 0 and world > 1 else None
        self.landing = torch.zeros(self.slab_bytes, dtype=torch.uint8, device=device) if stage_on_cpu and rank == 0 else None

    def gather(self, fb, out=None):
        import torch.distributed as dist
        torch = self.torch
        stream = torch.cuda.current_stream().cuda_stream
        self.scene.pack_tiles_device(fb.data_ptr(), self.width, self.height, self.fmt, self.rank, self.world, self.slab.data_ptr(), stream)
        if self.stage_on_cpu:
            self.xslab.copy_(self.slab)
        if self.world > 1:
            dist.gather(self.xslab, self.recv if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        if out is None:
            out = torch.zeros((self.height, self.width, 3), dtype=torch.uint8 if self.fmt else torch.float32, device=fb.device)
        for r in range(self.world):
            src = self.recv[r] if self.world > 1 else self.xslab
            if self.stage_on_cpu:
                self.landing.copy_(src)
                src = self.landing
            self.scene.unpack_tiles_device(src.data_ptr(), self.width, self.height, self.fmt, r, self.world, out.data_ptr(), stream)
            if self.stage_on_cpu:
                torch.cuda.current_stream().synchronize()      # (the one landing buffer is reused for the next rank's slab)
        return out
