"""Multi-GPU host side: one process per GPU, 8x8-pixel tiles dealt round-robin, and the ONE
exchange step of the path — the final framebuffer gather to rank 0 (RCCL over xGMI when the
tensors live on MI355X, gloo on CPU in tests).  No all-reduce: every pixel is produced by
exactly one rank, so the collective is a gather of equal-sized tile slabs.
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
