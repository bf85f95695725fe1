"""The N>1 host path: two gloo ranks each render their round-robin tile partition and the framebuffer
gather of vecchio_amd/distributed.py must reproduce the single-rank image bit for bit (tiles are
independent and the RNG is keyed per pixel, so it has to be exact).  On the CPU the oracle stands in for
the renderer (it honours the same tile_rank/tile_world contract); the -m gpu variant drives the HIP path
through the C ABI from both ranks (sharing the box's one MI355X), f32 and RGB8 slabs."""
import os
import sys

import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, outdir, use_hip):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vecchio_amd import DeviceScene, HostScene, ffi
    from vecchio_amd.distributed import FramebufferGather
    hs = HostScene("cornell_box", 1)
    cam = hs.next_camera()
    p = hs.params(w, 4, 20, height=h, tile_rank=rank, tile_world=world)
    if use_hip:
        ds = DeviceScene(hs.desc, device=0)           # both ranks share the one GPU of the box
        img, _ = ds.render(cam, p)
        p8 = hs.params(w, 4, 20, height=h, tile_rank=rank, tile_world=world, output_format=ffi.VK_OUTPUT_RGB8)
        img8, _ = ds.render(cam, p8)
        ds.close()
    else:
        import oracle_ffi as O
        img, _ = O.render(hs.desc, cam, p, threads=2)
    fb = torch.from_numpy(img.copy())
    g = FramebufferGather(w, h, rank, world, "cpu")
    full = g.gather(fb)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), full.numpy())
    if use_hip:
        g8 = FramebufferGather(w, h, rank, world, "cpu", rgb8=True)
        full8 = g8.gather(torch.from_numpy(img8.copy()))
        dist.barrier()
        if rank == 0:
            np.save(os.path.join(outdir, "gathered8.npy"), full8.numpy())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_gather_of_the_hip_path_is_bit_exact(device, host_scenes, tmp_path):
    """two processes, each calling vk_render for its tile partition on the MI355X, gathered over gloo"""
    import golden_checks as G
    from vecchio_amd import DeviceScene
    w, h = 52, 44
    mp.spawn(_worker, args=(2, free_port(), w, h, str(tmp_path), True), nprocs=2, join=True)
    hs, cam = host_scenes("cornell_box")
    ds = DeviceScene(hs.desc)
    single, _ = ds.render(cam, hs.params(w, 4, 20, height=h))
    ds.close()
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), single)
    assert np.array_equal(np.load(tmp_path / "gathered8.npy"), G.to_color(single)[::-1])


def test_two_rank_gather_is_bit_exact(oracle, host_scenes, tmp_path):
    w, h = 52, 44          # not multiples of 8: exercises edge tiles
    mp.spawn(_worker, args=(2, free_port(), w, h, str(tmp_path), False), nprocs=2, join=True)
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
        top = np.concatenate([tile_pixel_indices(w, h, r, world, top_down=True) for r in range(world)])
        assert np.array_equal(top, (h - 1 - allidx // w) * w + allidx % w)      # RGB8 images are stored top row first
