"""The N>1 host path on CPU: two gloo ranks each render their round-robin tile partition (the
oracle stands in for the GPU renderer here — it honours the same tile_rank/tile_world contract)
and the framebuffer gather of vecchio_amd/distributed.py must reproduce the single-rank image
bit for bit (tiles are independent and the RNG is keyed per pixel, so it has to be exact)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi as O
    from vecchio_amd import HostScene
    from vecchio_amd.distributed import FramebufferGather
    hs = HostScene("cornell_box", 1)
    cam = hs.next_camera()
    p = hs.params(w, 4, 20, height=h, tile_rank=rank, tile_world=world)
    img, _ = O.render(hs.desc, cam, p, threads=2)
    fb = torch.from_numpy(img.copy())
    g = FramebufferGather(w, h, rank, world, "cpu")
    full = g.gather(fb)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), full.numpy())
    dist.destroy_process_group()


def test_two_rank_gather_is_bit_exact(oracle, host_scenes, tmp_path):
    w, h = 52, 44          # not multiples of 8: exercises edge tiles
    mp.spawn(_worker, args=(2, 29517, w, h, str(tmp_path)), nprocs=2, join=True)
    gathered = np.load(tmp_path / "gathered.npy")
    hs, cam = host_scenes("cornell_box")
    p = hs.params(w, 4, 20, height=h)
    single, _ = oracle.render(hs.desc, cam, p, threads=4)
    assert np.array_equal(gathered, single)


def test_tile_partition_covers_image_once():
    from vecchio_amd.distributed import tile_pixel_indices
    for (w, h, world) in ((52, 44, 2), (64, 64, 8), (17, 9, 3)):
        allidx = np.concatenate([tile_pixel_indices(w, h, r, world) for r in range(world)])
        assert len(allidx) == w * h and len(np.unique(allidx)) == w * h
