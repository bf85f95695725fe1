#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel sample loop on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch = one vk_render_device() of the workload:
BASELINE config C2, InOneWeekend random-spheres scene, 1920x1080, 1024 spp, max_depth 50
(sphere-only BVH megakernel), scene resident in HBM/LDS before the timed region.  With N > 1
(one process per GPU, torch.distributed/RCCL) the image's 8x8 tiles are dealt round-robin over
the ranks (fixed total work => "strong" scaling) and the step ends with the framebuffer gather
to rank 0 — the path's only exchange.

Also reported in the same JSON line:
  roofline     algorithmic bytes per launch (SURVEY §8d formula, visit counts from the oracle's
               counters on a bounded sample of the same scene/seed) / HIP-event time of the launch
  cpu_baseline the oracle (restated CPU path, "port") timed on this box's host cores on a bounded
               sample of the same workload (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (scene builder, width, spp, max_depth, label)
    "C2": ("random_spheres_iow", 1920, 1024, 50, "C2: InOneWeekend random spheres 1920x1080, 1024 spp, depth 50"),
    "C3": ("final_scene", 800, 10000, 50, "C3: TheNextWeek final scene 800x800, 10000 spp, depth 50"),
    "C4": ("cornell_box", 1024, 4096, 50, "C4: Cornell box 1024x1024, 4096 spp, depth 50"),
    "C5": ("stress_spheres:500", 4096, 256, 50, "C5: 1M spheres 4096x4096, 256 spp, depth 50"),
}


def algorithmic_bytes_per_sample(c, spp):
    """SURVEY §8(d) / BASELINE.md §4 with the canonical record sizes."""
    n = float(c["samples"])
    return (32.0 * c["n_aabb"] + 16.0 * c["n_sphere"] + 36.0 * c["n_moving"] + 24.0 * c["n_rect"] + 32.0 * c["n_xform"] +
            8.0 * c["n_medium"] + 16.0 * c["n_closest"] + 3.0 * c["n_texel"] + 8.0 * 24.0 * c["n_perlin"]) / n + 12.0 / spp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--bvh", default="reference", choices=["reference", "sah"],
                    help="host-side BVH builder: the reference's BVHNode::new (default; what a drop-in host hands over) "
                         "or the host mirror's SAH builder over the same objects (SURVEY 8f-2)")
    ap.add_argument("--spp", type=int, default=0, help="override spp (marks the result as non-headline)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU sample (0 = size it to ~15 s of CPU work)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from vecchio_amd import DeviceScene, HostScene
    from vecchio_amd.distributed import FramebufferGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    # VK_BENCH_REHEARSAL=1: run the N-rank code path on ONE GPU (all ranks share device 0, gloo
    # instead of RCCL, tile slabs staged through host memory) - for checking the multi-process
    # logic on a single-GPU box; its numbers are not a scaling measurement.
    rehearsal = os.environ.get("VK_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" IS RCCL on ROCm

    scene_name, width, spp, depth, label = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    if args.bvh == "sah":
        scene_name += "+sah"
        label += " [SAH BVH over the same objects]"
    hs = HostScene(scene_name, 1)                        # scene seed 1
    cam = hs.next_camera()
    params = hs.params(width, spp, depth, seed=2, tile_rank=rank, tile_world=world)   # render seed 2
    height = params.height
    ds = DeviceScene(hs.desc, device=dev_index)          # scene upload: outside the timed region
    info = ds.info()
    fb = torch.zeros((height, width, 3), dtype=torch.float32, device=dev)
    full = torch.zeros_like(fb) if (world > 1 and rank == 0) else None
    gather = FramebufferGather(width, height, rank, world, dev, stage_on_cpu=rehearsal)
    stream = torch.cuda.current_stream().cuda_stream

    lib = ds._lib
    import ctypes as C
    lib.vk_scene_last_kernel_ms.restype = C.c_int
    lib.vk_scene_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]

    kernel_ms = []

    def step(record=False):
        st = ds.render_device(cam, params, fb.data_ptr(), stream)
        if world > 1:
            gather.gather(fb, full)
        if record:
            ms = C.c_double()
            if lib.vk_scene_last_kernel_ms(ds._h, C.byref(ms)) == 0:   # waits for this step's end event
                kernel_ms.append(ms.value)
        return st

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local_samples = 0
    for _ in range(args.steps):
        st = step(record=True)
        local_samples = st.samples
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rehearsal and world > 1 and rank == 0:
        p1 = hs.params(width, spp, depth, seed=2)
        ref, _ = ds.render(cam, p1)
        same = bool((full.cpu().numpy() == ref).all())
        print(f"rehearsal: gathered {world}-rank image bit-identical to 1-rank render: {same}", file=sys.stderr)
        assert same
    total_samples = width * height * spp                  # all ranks together, per step
    if rank == 0:
        value = total_samples * args.steps / elapsed / 1e6
        # ---- bounded oracle sample: visit counters (algorithmic bytes) + CPU baseline
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_ffi as O
        cores = os.cpu_count() or 1
        cpu = None
        cw = width if args.workload != "C5" else 1024     # C5's 16.7M pixels: sample a quarter-res grid on the CPU
        want_cpu = world == 1 and not args.no_cpu
        # probe pass (also gives the visit counters); the timed CPU sample is then sized to ~15 s
        pc = hs.params(cw, 1, depth, seed=2)
        tc0 = time.perf_counter()
        _, cnt = O.render(hs.desc, cam, pc, threads=cores)
        tc = time.perf_counter() - tc0
        cpu_spp = 1
        if want_cpu:
            rate = cnt.samples / tc
            cpu_spp = args.cpu_spp if args.cpu_spp > 0 else int(min(64, max(2, 15.0 * rate / (cw * pc.height))))
            pc = hs.params(cw, cpu_spp, depth, seed=2)
            tc0 = time.perf_counter()
            _, cnt = O.render(hs.desc, cam, pc, threads=cores)
            tc = time.perf_counter() - tc0
        c = cnt.as_dict()
        bps = algorithmic_bytes_per_sample(c, spp)
        if world == 1 and not args.no_cpu:
            cpu = {"value": round(c["samples"] / tc / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": f"{cw}x{pc.height} px x {cpu_spp} spp = {c['samples']} samples of the same scene/seed/depth, "
                             f"{tc:.1f} s, oracle (recursive CPU restatement) on {cores} threads"}
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else None
        # measured HBM traffic per launch: PMC counters cannot be collected from inside this process, so the
        # figure comes from the committed rocprofv3 passes of this same command (profiles/, tests/prof_r01.sh):
        # WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE counts half the bytes of wide reads -> upper bound)
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r01", f"{args.workload.lower()}_pmc_summary.json")
        if world == 1 and not args.spp and os.path.exists(prof):
            try:
                d = json.load(open(prof))["derived"]
                traffic = float(d["hbm_write_bytes_per_dispatch"] + d["hbm_read_bytes_per_dispatch"]["with_gfx950_x2_correction_upper_bound"])
            except Exception:
                traffic = None
        roof = None
        if k_ms:
            achieved = bps * local_samples / (k_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes per launch from the committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command (profiles/r01/), not re-measured in this run",
                    "algorithmic_bytes_per_sample": round(bps, 1), "kernel_ms": round(k_ms, 3),
                    "note": "algorithmic bytes (SURVEY 8d record sizes x oracle visit counts); the scene is LDS/L2 resident, "
                            "so measured HBM traffic is far below this (see profiles/)"}
        out = {
            "metric": "Msamples/sec (pixels x spp)", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {"workload": label if not args.spp else label + f" [spp overridden to {spp}]", "scene_seed": 1, "render_seed": 2,
                       "integrator": "scatter" if hs.integrator else "pdf", "tiles": "8x8 round-robin over ranks",
                       "bvh_builder": args.bvh, "bvh_items": info.n_items, "scene_in_lds": bool(info.lds_bytes)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ds.close()


if __name__ == "__main__":
    main()
