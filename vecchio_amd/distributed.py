"""Multi-GPU host side: one process per GPU, 8x8-pixel tiles dealt round-robin, and the ONE
exchange step of the path — the final framebuffer gather to rank 0 (RCCL over xGMI when the
tensors live on MI355X, gloo on CPU in tests).  No all-reduce: every pixel is produced by
exactly one rank, so the collective is a gather of equal-sized tile slabs.
"""
import numpy as np

TILE = 8


def tile_pixel_indices(width, height, rank, world):
    """Flat pixel indices (y*width + x, y = 0 bottom) of the tiles owned by `rank`, tile by tile
    in the order the kernel deals them (tile t -> rank t % world)."""
    tiles_x = (width + TILE - 1) // TILE
    tiles_y = (height + TILE - 1) // TILE
    out = []
    ys, xs = np.mgrid[0:TILE, 0:TILE]
    for t in range(rank, tiles_x * tiles_y, world):
        tx, ty = (t % tiles_x) * TILE, (t // tiles_x) * TILE
        x, y = (tx + xs).ravel(), (ty + ys).ravel()
        ok = (x < width) & (y < height)
        out.append((y[ok] * width + x[ok]).astype(np.int64))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


class FramebufferGather:
    """Pre-computes the per-rank index sets once; `gather(fb)` moves each rank's tile slab to
    rank 0 with a single torch.distributed.gather and scatters it into the full image."""

    def __init__(self, width, height, rank, world, device, stage_on_cpu=False):
        """stage_on_cpu: exchange through host tensors (gloo rehearsals: gloo cannot gather GPU tensors)."""
        import torch
        self.torch = torch
        self.stage_on_cpu = stage_on_cpu
        self.width, self.height, self.rank, self.world = width, height, rank, world
        idx = [tile_pixel_indices(width, height, r, world) for r in range(world)]
        self.slab_len = max(len(i) for i in idx)
        self.own = torch.from_numpy(idx[rank]).to(device)
        self.all_idx = [torch.from_numpy(i).to(device) for i in idx] if rank == 0 else None
        xdev = "cpu" if stage_on_cpu else device
        self.slab = torch.zeros((self.slab_len, 3), dtype=torch.float32, device=xdev)
        self.recv = [torch.zeros((self.slab_len, 3), dtype=torch.float32, device=xdev) for _ in range(world)] if rank == 0 else None

    def gather(self, fb, out=None):
        """fb: (height, width, 3) float32 tensor holding this rank's tiles.  Returns the full image on
        rank 0 (written into `out` if given), None elsewhere."""
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
