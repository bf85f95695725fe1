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


def _device_worker(rank, world, port, w, h, outdir):
    """the product's exchange: render on the device, pack the slab with the library's tile kernel, ONE gather, unpack on rank 0 — no
    torch indexing anywhere (gloo cannot move GPU tensors: the slabs are staged through host memory, as in bench.py's rehearsal)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vecchio_amd import DeviceScene, HostScene
    from vecchio_amd.distributed import DeviceFramebufferGather
    hs = HostScene("cornell_box", 1)
    cam = hs.next_camera()
    p = hs.params(w, 4, 20, height=h, tile_rank=rank, tile_world=world)
    ds = DeviceScene(hs.desc, device=0)
    dev = torch.device("cuda", 0)
    fb = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    for rgb8 in (False, True):
        g = DeviceFramebufferGather(ds, w, h, rank, world, dev, stage_on_cpu=True, rgb8=rgb8)
        ds.render_device(cam, p, fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
        full = g.gather(fb)
        dist.barrier()
        if rank == 0:
            np.save(os.path.join(outdir, "dev_gathered8.npy" if rgb8 else "dev_gathered.npy"), full.cpu().numpy())
    ds.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_device_gather_packs_and_unpacks_on_the_device(device, host_scenes, tmp_path):
    import golden_checks as G
    from vecchio_amd import DeviceScene
    w, h = 52, 44
    mp.spawn(_device_worker, args=(2, free_port(), w, h, str(tmp_path)), nprocs=2, join=True)
    hs, cam = host_scenes("cornell_box")
    ds = DeviceScene(hs.desc)
    single, _ = ds.render(cam, hs.params(w, 4, 20, height=h))
    ds.close()
    assert np.array_equal(np.load(tmp_path / "dev_gathered.npy"), single)
    assert np.array_equal(np.load(tmp_path / "dev_gathered8.npy"), G.to_color(single)[::-1])
    import inspect
    from vecchio_amd import distributed
    src = inspect.getsource(distributed.DeviceFramebufferGather)
    assert "index_select" not in src and "index_copy" not in src


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


@pytest.mark.gpu
def test_bench_gpus_n_without_a_launcher_starts_n_ranks(device):
    """`python bench.py --gpus 2` — the driver's command shape, no torchrun — must start two ranks itself (as a CHILD process of
    a parent that never touches the GPU), render the frame tile-parallel, gather it and say so in its JSON line: world size,
    backend, one device record per rank, and the gathered image bit-identical to ONE device's render of the whole frame.
    Rehearsal on the box's single GPU: both ranks on device 0, gloo instead of RCCL (VK_BENCH_REHEARSAL=1)."""
    import json
    import subprocess
    env = dict(os.environ, VK_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "16",
                        "--no-cpu", "--no-also", "--no-traffic"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    # ... and with the output stage fused into the exchange (RGB8 slabs)
    r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "16",
                         "--no-cpu", "--no-also", "--no-traffic", "--rgb8"], env=env, capture_output=True, text=True, timeout=600)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-4000:]
    d8 = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][-1])
    assert d8["distributed"]["gathered_image_equals_one_gpu_render"] is True and d8["verified"]["ok"] is True
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    dd = d["distributed"]
    assert dd["world_size"] == 2 and dd["self_launched"] and len(dd["ranks"]) == 2 and {x["rank"] for x in dd["ranks"]} == {0, 1}
    assert dd["gathered_image_equals_one_gpu_render"] is True
    assert d["verified"]["ok"] is True
