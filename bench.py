#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel sample loop on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch = one vk_render_device() of the workload:
BASELINE config C2, InOneWeekend random-spheres scene, 1920x1080, 1024 spp, max_depth 50
(sphere-only BVH megakernel), scene resident in HBM/LDS before the timed region.  With N > 1
(one process per GPU, torch.distributed/RCCL) the image's 8x8 tiles are dealt round-robin over
the ranks (fixed total work => "strong" scaling) and the step ends with the framebuffer gather
to rank 0 — the path's only exchange.

Also reported in the same JSON line:
  verified     the framebuffer of the LAST TIMED STEP, compared on a sparse tile subset (all over the frame, at
               the full spp) with the oracle's render of the same pixels: pixels checked, max |dRGB|
  roofline     algorithmic bytes per launch (SURVEY §8d formula, visit counts from the oracle's counters on a
               bounded sample of the same scene/seed) / HIP-event time of the launch; plus `issue`: the
               `traffic` = the HBM bytes of one launch and the instruction counts of `issue`, measured in this run by three
               one-step child runs under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE, SQ counters; --no-traffic or a profiler that
               cannot run: the committed profile's figures); `issue` = the physical bound of this kernel — VALU issue slots (wave-instructions per sample from the committed
               rocprofv3 PMC pass of this command, 2 cycles each on a SIMD-32, 1024 SIMDs x 2.4 GHz)
  cpu_baseline the oracle (restated CPU path, "port") timed on this box's host cores on a bounded sample of the
               same workload (N = 1 only)
  handed_over_tree  one step of the same workload with VK_SCENE_REFERENCE_TREE (the tree of the description, nothing rebuilt), verified
               the same way: what the rebuilt tree of exact re-treeing is worth (N = 1 only)
  config.also  one step each of the other BASELINE configs at full size (C4, C3, C5 — C5 on the tree as handed over, as the default
               walks it (the near form of exact re-treeing) AND with VK_SCENE_EMPIRICAL_TREES), of C2 in the empirical form and with
               VK_SCENE_FAST_ACCEL, each verified the same way and, on a rebuilt tree, compared bit for bit with the same workload's frame
               on the tree as handed over (N = 1 only; --no-also skips them)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2.0   # 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles, 2.4 GHz max clock
PROFILE_ROUND = "r05"

WORKLOADS = {
    # name: (scene builder, width, spp, max_depth, label)
    "C2": ("random_spheres_iow", 1920, 1024, 50, "C2: InOneWeekend random spheres 1920x1080, 1024 spp, depth 50"),
    "C3": ("final_scene", 800, 10000, 50, "C3: TheNextWeek final scene 800x800, 10000 spp, depth 50"),
    "C4": ("cornell_box", 1024, 4096, 50, "C4: Cornell box 1024x1024, 4096 spp, depth 50"),
    "C5": ("stress_spheres:500", 4096, 256, 50, "C5: 1M spheres 4096x4096, 256 spp, depth 50"),
}



def is_main_kernel(name):
    """a render_kernel<F, LDS_SCENE, MINW, STATS, COST, GRID> instance that is not the probe launch's (COST = true)"""
    import re
    m = re.search(r"render_kernel<\s*\d+u?,\s*(?:true|false),\s*\d+,\s*(?:true|false),\s*(true|false)(?:,\s*(?:true|false))?\s*>", name)
    return bool(m) and m.group(1) == "false"


def WORKLOAD_SAMPLES(name):
    _, w, spp, _, _ = WORKLOADS[name]
    h = {"C2": 1080, "C3": 800, "C4": 1024, "C5": 4096}[name]
    return w * h * spp


def algorithmic_bytes_per_sample(c, spp, walked_only=False):
    """SURVEY §8(d) / BASELINE.md §4 with the canonical record sizes.  `walked_only` leaves out the visits of segments whose
    ray is NaN / infinite: the reference walks the whole tree for such a ray (every box passes, every sphere fails) and
    the device skips that walk in sphere-only scenes, where its outcome is known (vk_trace.h begin_segment)."""
    n = float(c["samples"])
    aabb, sph = c["n_aabb"], c["n_sphere"]
    if walked_only:
        aabb, sph = aabb - c["n_aabb_nonfinite"], sph - c["n_sphere_nonfinite"]
    return (32.0 * aabb + 16.0 * sph + 36.0 * c["n_moving"] + 24.0 * c["n_rect"] + 32.0 * c["n_xform"] +
            8.0 * c["n_medium"] + 16.0 * c["n_closest"] + 3.0 * c["n_texel"] + 8.0 * 24.0 * c["n_perlin"]) / n + 12.0 / spp


LDS_PEAK_BPS = 150e12           # MI355X_MICROARCH.md: ~150 TB/s aggregate for ds_read_b64/b128 with every CU streaming
L2_PEAK_BPS = 34.5e12           # MI355X_MICROARCH.md: ~34.5 TB/s aggregate L2


def physical_bounds(c, samples, seconds, scene_in_lds, issue, walked=None):
    """What physically limits the kernel, every fraction <= 1 by construction (unlike the algorithmic-HBM `roofline`, whose bytes
    are served from LDS / L2):
      valu_issue  vector instruction issue: wave-instructions/s against 1024 SIMDs x 2.4 GHz / 2 cycles; x lane fill = the share of
                  the chip's 78.6 T lane-instructions/s that does useful work
      lds         (scene staged in LDS) the traversal's record bytes — 32 B per box test, 16 B per sphere test, oracle visit counts —
                  against the LDS arrays' ~150 TB/s
      l2          (scene read from global memory) the same record bytes as an upper bound of the L2 request traffic (what L1 absorbs
                  is not subtracted) against ~34.5 TB/s
    The dominant one is named in `bound`."""
    n = float(c["samples"])
    rec = 32.0 * (c["n_aabb"] - c["n_aabb_nonfinite"]) + 16.0 * (c["n_sphere"] - c["n_sphere_nonfinite"]) + 36.0 * c["n_moving"] + 24.0 * c["n_rect"] + \
        32.0 * c["n_xform"] + 8.0 * c["n_medium"]
    note = "record bytes from the oracle's visit counts: the device walks the same tree, item for item"
    if walked and "sphere_tests" not in walked:
        # the tree as handed over, but for the second call of `len == 1` nodes over draw-free instances, which the device does not make
        # (vk_linearize.cpp draw_free_instance): its box tests counted by the kernel's per-lane code on the host, the primitive visits
        # still the oracle's (an upper bound)
        rec -= 32.0 * (walked["oracle_box_tests"] - walked["box_tests"]) * n
        note = ("box tests of the walk the device performs (tests/emu: the tree as handed over without the second call of len-1 nodes over "
                "draw-free instances); primitive visits from the oracle's counts (upper bound)")
    elif walked:      # the device walks a rebuilt tree: its own visits (tests/emu, the kernel's per-lane code on the host, same sample)
        rec = (32.0 * walked["box_tests"] + 16.0 * walked["sphere_tests"]) * n
        note = ("record bytes of the walk the device performs (rebuilt tree, second walks and requeued samples included), counted by the "
                "kernel's per-lane code built for the host (tests/emu) on the same bounded sample as the oracle's counters")
    rate = rec / n * samples / seconds
    out = {"bound": "valu_issue", "unit": "wave-instr/s", "achieved": None, "peak": VALU_WAVE_INSTR_PER_S, "frac": None,
           "lane_fill": None, "useful_lane_frac": None,
           "memory": {"level": "lds" if scene_in_lds else "l2", "record_bytes_per_sample": round(rec / n, 1), "achieved_Bps": round(rate, 1),
                      "peak_Bps": LDS_PEAK_BPS if scene_in_lds else L2_PEAK_BPS, "frac": round(rate / (LDS_PEAK_BPS if scene_in_lds else L2_PEAK_BPS), 4),
                      "note": note}}
    if issue:
        out.update(achieved=issue["achieved_wave_instr_per_s"], frac=issue["frac"], lane_fill=issue["lane_fill"],
                   useful_lane_frac=round(issue["frac"] * issue["lane_fill"], 4), counters=issue.get("note"))
    if out["frac"] is not None and out["memory"]["frac"] > out["frac"]:
        out["bound"] = out["memory"]["level"]
    return out


def pmc_summary(workload):
    """the committed rocprofv3 PMC summary of this workload's bench command (profiles/<round>/), or None"""
    for rnd in (PROFILE_ROUND, "r04", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, f"{workload.lower()}_pmc_summary.json")
        if os.path.exists(path) and os.path.getsize(path) > 0:
            try:
                return json.load(open(path)), os.path.relpath(path, ROOT)
            except Exception:
                pass
    return None, None


def measure_counters_live(workload, timeout_s=75):
    """Hardware counters of ONE launch of this workload's production kernel, measured now: child runs of this script (one step each)
    under rocprofv3 --pmc, one pass per counter group as MI355X_MICROARCH.md prescribes (kernel trace only, the program directly
    after `--`): FETCH_SIZE, WRITE_SIZE (KiB; on gfx950 FETCH_SIZE counts half the bytes of wide reads, so the read figure is given
    as the x2 upper bound) and the SQ instruction counters behind `roofline.issue`.  Returns (dict, None) or (None, reason): the
    caller then falls back to the committed profile."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "this run is itself being profiled"
    got = {}
    for group in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"),
                  ("SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"), ("TCC_WRITE_sum", "TCC_ATOMIC_sum")):
        tmp = tempfile.mkdtemp(prefix="vk_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu", "--no-verify", "--no-also", "--no-traffic"]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            vals = {c: [] for c in group}
            for f in glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if is_main_kernel(r["Kernel_Name"]) and r["Counter_Name"] in vals:
                        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c in group:
                if not vals[c]:
                    return None, f"no {c} row for the render kernel"
                got[c] = sum(vals[c])        # ONE step was run: the frame's production dispatches (two when the library launches 16 + 12 waves per CU)
        except Exception as e:      # a profiler that cannot run here must not take the benchmark down
            return None, f"{group[0]} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    got["read_bytes"], got["read_bytes_upper_bound"], got["written_bytes"] = got["FETCH_SIZE"] * 1024.0, 2.0 * got["FETCH_SIZE"] * 1024.0, got["WRITE_SIZE"] * 1024.0
    return got, None


def verify_against_oracle(O, hs, cam, img, width, spp, depth, budget_samples=1.5e6, min_pixels=1024):
    """Compares a sparse subset of 8x8 tiles of `img` (numpy, (h, w, 3), y up) with the oracle's render of exactly
    those pixels at the same seed and the FULL spp.  Every k-th tile (k prime, so the subset wanders over the
    whole frame) such that the oracle traces about `budget_samples` samples."""
    import numpy as np
    height = img.shape[0]
    tiles = ((width + 7) // 8) * ((height + 7) // 8)
    # (at least min_pixels pixels whatever the spp: C3's 10 000 spp left the budget 128 pixels)
    want_tiles = max(2, (min_pixels + 63) // 64, int(budget_samples / (64.0 * spp)))
    k = max(1, tiles // want_tiles)
    while k > 1 and any(k % d == 0 for d in range(2, int(k ** 0.5) + 1)):
        k += 1
    po = hs.params(width, spp, depth, seed=2, height=height, tile_rank=k // 2, tile_world=k)
    ref = np.full((height, width, 3), -1.0, np.float32)
    t0 = time.perf_counter()
    st = O.load().oracle_render(hs.desc, C.byref(cam), C.byref(po), ref.ctypes.data, min(usable_cpus(), 64), None)
    if st != 0:
        return {"error": f"oracle status {st}"}
    mask = ref[..., 0] >= 0
    err = float(np.abs(img[mask] - ref[mask]).max())
    return {"pixels": int(mask.sum()), "tiles": f"every {k}th 8x8 tile", "spp": spp, "max_abs_err": err, "tolerance": 1e-4,
            "ok": bool(err < 1e-4 and np.isfinite(img).all()), "oracle_seconds": round(time.perf_counter() - t0, 2)}


def usable_cpus():
    """CPUs this process may really use: its affinity mask, capped by the cgroup's CPU quota (os.cpu_count() is the machine's)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def self_launch(n):
    """torch.distributed.run with n ranks of this script and this command line, as a child process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ, VK_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # relay: the ranks' JSON line goes to stdout, everything else they print there (gloo's connection banners) to stderr, so that
    # stdout holds exactly the one line the contract asks for
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


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
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run check of the timed framebuffer against the oracle")
    ap.add_argument("--no-also", action="store_true", help="skip the single steps of the other BASELINE configs")
    ap.add_argument("--reference-tree", action="store_true", help="set VK_SCENE_REFERENCE_TREE: walk only the tree handed over (no exact "
                    "re-treeing of scenes of spheres only)")
    ap.add_argument("--empirical-trees", action="store_true", help="set VK_SCENE_EMPIRICAL_TREES: rebuilt trees also where their exactness is "
                    "measured, not proven (include/vecchio_amd.h)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure the HBM traffic of a launch now (two one-step child runs under rocprofv3 --pmc); "
                         "roofline.traffic then comes from the committed profile")
    ap.add_argument("--fast-accel", action="store_true",
                    help="set VK_SCENE_FAST_ACCEL in the scene description: the library rebuilds draw-free subtrees with its SAH builder "
                         "(opt-in; not the headline: see include/vecchio_amd.h)")
    ap.add_argument("--rgb8", action="store_true", help="N ranks: gather RGB8 slabs (Vec3::to_color fused into the pack, vec3.rs:54-61) "
                    "instead of f32 ones")
    ap.add_argument("--rccl-gather", action="store_true", help="--in-library: set VK_SCENE_RCCL_GATHER (the library moves the tile slabs by "
                    "grouped ncclSend / ncclRecv instead of hipMemcpyPeerAsync)")
    ap.add_argument("--in-library", action="store_true",
                    help="N GPUs from ONE process through vk_scene_create_multi (the library deals tiles, gathers on device 0) "
                         "instead of one process per GPU; run without torchrun")
    args = ap.parse_args()

    # `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU, as a CHILD process —
    # never exec: nothing in this process has touched the GPU yet, and nothing will — relay its output and exit with its code.
    if args.gpus > 1 and "RANK" not in os.environ and not args.in_library:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from vecchio_amd import DeviceScene, HostScene
    from vecchio_amd.distributed import DeviceFramebufferGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.in_library:
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
    n_gpus = args.gpus if args.in_library else world
    if args.in_library:
        assert world == 1, "--in-library is one process driving N devices"
    in_lib_devices = None
    if args.in_library:
        n_vis = torch.cuda.device_count()
        in_lib_devices = [i % n_vis for i in range(args.gpus)] if rehearsal else list(range(args.gpus))

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    cores = usable_cpus()          # threads for the oracle: affinity mask capped by the cgroup quota

    TREE_NAMES = {0: "handed over", 1: "rebuilt, proven (exact re-treeing with grown gates)", 2: "rebuilt, empirical (VK_SCENE_EMPIRICAL_TREES)",
                  3: "rebuilt object by object (VK_SCENE_FAST_ACCEL)",
                  4: "rebuilt, proven (exact re-treeing, near form: own-box gates + reach / clearance, failed segments walked again as handed over)",
                  5: "no tree: a grid over the layer of small spheres, proven (exact re-treeing, grid form: every sphere that can hold a "
                     "candidate is tested, failed segments' samples rendered again as handed over)"}

    def run_workload(name, steps, warmup, spp_override=0, want_cpu=False, bvh="reference", fast_accel=False, live_traffic=False,
                     handed_over_tree=False, empirical=False):
        scene_name, width, spp, depth, label = WORKLOADS[name]
        if spp_override:
            spp = spp_override
            label += f" [spp overridden to {spp}]"
        if bvh == "sah":
            scene_name += "+sah"
            label += " [SAH BVH over the same objects]"
        hs = HostScene(scene_name, 1)                        # scene seed 1
        if fast_accel:
            from vecchio_amd import ffi
            hs.desc.contents.flags = ffi.VK_SCENE_FAST_ACCEL
            label += " [VK_SCENE_FAST_ACCEL: draw-free subtrees rebuilt by the library]"
        if handed_over_tree:
            # the default for a scene of spheres only is exact re-treeing (vk_trace.h segment_unsafe); this walks the tree as built by
            # the host mirror of BVHNode::new and nothing else
            from vecchio_amd import ffi
            hs.desc.contents.flags = ffi.VK_SCENE_REFERENCE_TREE
            label += " [VK_SCENE_REFERENCE_TREE: only the tree handed over]"
        if empirical:
            # rebuilt trees also where their exactness is measured, not proven (include/vecchio_amd.h)
            from vecchio_amd import ffi
            hs.desc.contents.flags = ffi.VK_SCENE_EMPIRICAL_TREES
            label += " [VK_SCENE_EMPIRICAL_TREES]"
        if args.in_library and args.rccl_gather:
            from vecchio_amd import ffi
            hs.desc.contents.flags |= ffi.VK_SCENE_RCCL_GATHER
            label += " [VK_SCENE_RCCL_GATHER]"
        cam = hs.next_camera()
        params = hs.params(width, spp, depth, seed=2, tile_rank=rank, tile_world=world)   # render seed 2
        height = params.height
        # scene upload: outside the timed region.  (The empirical leg: VK_GATE_PROOF=0 makes the library prefer the empirical unit form to
        # the proven forms, which VK_SCENE_EMPIRICAL_TREES alone only allows where neither applies; read at scene creation.)
        if empirical:
            os.environ["VK_GATE_PROOF"] = "0"
        try:
            ds = DeviceScene(hs.desc, devices=in_lib_devices) if args.in_library else DeviceScene(hs.desc, device=dev_index)
        finally:
            os.environ.pop("VK_GATE_PROOF", None)
        info = ds.info()
        fb = torch.zeros((height, width, 3), dtype=torch.float32, device=dev)
        # the gathered frame on rank 0: f32 (y up), or with --rgb8 the reference's output stage fused into the exchange (bytes through
        # Vec3::to_color, top row first: a quarter of the traffic over xGMI)
        full = torch.zeros((height, width, 3), dtype=torch.uint8 if args.rgb8 else torch.float32, device=dev) if (world > 1 and rank == 0) else None
        gather = DeviceFramebufferGather(ds, width, height, rank, world, dev, stage_on_cpu=rehearsal, rgb8=args.rgb8) if world > 1 else None
        stream = torch.cuda.current_stream().cuda_stream
        kernel_ms = []

        def step(record=False):
            st = ds.render_device(cam, params, fb.data_ptr(), stream)
            if world > 1:
                gather.gather(fb, full)
            if record:
                kernel_ms.append(ds.last_kernel_ms())      # waits for this step's end event
            return st

        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        local_samples = 0
        for _ in range(steps):
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
        total_samples = width * height * spp                  # all ranks together, per step
        res = None
        if rank == 0:
            value = total_samples * steps / elapsed / 1e6
            final = (full if world > 1 else fb).cpu().numpy()    # the image the last timed step produced
            verified = None
            rgb8_gathered = world > 1 and args.rgb8
            if not args.no_verify and not rgb8_gathered:
                verified = verify_against_oracle(O, hs, cam, final, width, spp, depth)
            # ---- bounded oracle sample: visit counters (algorithmic bytes) + CPU baseline
            cw = width if name != "C5" else 1024     # C5's 16.7M pixels: sample a quarter-res grid on the CPU
            pc = hs.params(cw, 1, depth, seed=2)
            tc0 = time.perf_counter()
            _, cnt = O.render(hs.desc, cam, pc, threads=cores)
            tc = time.perf_counter() - tc0
            cpu = None
            if want_cpu:
                # a GPU box may show more CPUs than its share delivers (16 per GPU on this pool): the 1-spp pass above is also
                # timed on min(cores, 16) threads and the faster thread count is the one used and reported
                threads, rate = cores, cnt.samples / tc
                if cores > 16:
                    tc0 = time.perf_counter()
                    _, c16 = O.render(hs.desc, cam, pc, threads=16)
                    r16 = c16.samples / (time.perf_counter() - tc0)
                    if r16 > rate:
                        threads, rate = 16, r16
                cpu_spp = args.cpu_spp if args.cpu_spp > 0 else int(min(64, max(2, 15.0 * rate / (cw * pc.height))))
                pc = hs.params(cw, cpu_spp, depth, seed=2)
                tc0 = time.perf_counter()
                _, cnt = O.render(hs.desc, cam, pc, threads=threads)
                tc = time.perf_counter() - tc0
                cpu = {"value": round(cnt.samples / tc / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                       "per_thread": round(cnt.samples / tc / 1e6 / threads, 4), "usable_cpus": cores, "machine_cpus": os.cpu_count(),
                       "sample": f"{cw}x{pc.height} px x {cpu_spp} spp = {cnt.samples} samples of the same scene/seed/depth, "
                                 f"{tc:.1f} s, oracle (recursive CPU restatement, 8x8-tile stealing) on {threads} threads"}
            c = cnt.as_dict()
            bps = algorithmic_bytes_per_sample(c, spp)
            sphere_only = info.features == 0       # the sphere-only kernel variant: the one that skips those walks
            bps_walked = algorithmic_bytes_per_sample(c, spp, walked_only=sphere_only)
            # The walk the device PERFORMS.  On the tree as handed over it is the oracle's, item for item (minus the NaN rays' walks).  On a
            # rebuilt tree it is counted by the kernel's own per-lane code built for the host (tests/emu: same linearisation, same gates,
            # second walks and requeued samples included) on the same bounded sample.
            walked = None
            if info.tree != 0 and sphere_only:
                import emu_ffi
                os.environ["EMU_GLOBAL_VARIANT"] = "0" if info.lds_bytes else "1"
                # (the near form: C5's camera is 109 above the field, farther than `reach` from every sphere, so the library starts the
                # primary rays on the tree as handed over — vk_api.hip, DScene::primary_ref — and so does the count of its walk)
                os.environ["EMU_PRIMARY_REF"] = "1" if (info.tree == 4 and name == "C5") else "0"
                os.environ["VK_GRID_FORM"] = "1" if info.tree == 5 else "0"     # (the grid form only where the device walks it: scenes in LDS)
                if empirical:
                    os.environ["VK_GATE_PROOF"] = "0"
                emu_ffi.take_visit_counts()
                pe = hs.params(cw, 1, depth, seed=2)
                _, ps_e, _, _ = emu_ffi.render_samples(hs.desc, cam, pe, threads=cores)
                nb, ns = emu_ffi.take_visit_counts()
                os.environ.pop("VK_GATE_PROOF", None); os.environ.pop("VK_GRID_FORM", None)
                ne = float(ps_e.shape[0])
                walked = {"box_tests": nb / ne, "sphere_tests": ns / ne, "oracle_box_tests": c["n_aabb"] / float(c["samples"]),
                          "oracle_sphere_tests": c["n_sphere"] / float(c["samples"])}
                bps_walked = 32.0 * walked["box_tests"] + 16.0 * walked["sphere_tests"] + 16.0 * c["n_closest"] / float(c["samples"]) + 12.0 / spp
            if walked is None and (info.features & 0x10):       # scenes with instances: see physical_bounds
                import emu_ffi
                emu_ffi.take_visit_counts()
                pe = hs.params(cw, 1, depth, seed=2)
                _, ps_e, _, _ = emu_ffi.render_samples(hs.desc, cam, pe, threads=cores)
                nb, _ns = emu_ffi.take_visit_counts()
                walked = {"box_tests": nb / float(ps_e.shape[0]), "oracle_box_tests": c["n_aabb"] / float(c["samples"])}
                bps_walked = bps - 32.0 * (walked["oracle_box_tests"] - walked["box_tests"])
            k_ms = float(np.mean(kernel_ms)) if kernel_ms else None
            # measured HBM traffic and instruction counts per launch: PMC counters cannot be collected from inside this
            # process, so they come from the committed rocprofv3 passes of this same command (profiles/, tools/experiments/prof_r02.sh):
            # WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE counts half the bytes of wide reads -> upper bound)
            traffic, issue, prof_path = None, None, None
            t_read = t_write = scratch_share = scratch_detail = None
            if n_gpus == 1 and not spp_override and bvh == "reference" and not fast_accel and not handed_over_tree and not empirical:
                prof, prof_path = pmc_summary(name)
                if prof:
                    d = prof.get("derived", {})
                    try:
                        t_write = float(d["hbm_write_bytes_per_dispatch"])
                        t_read = float(d["hbm_read_bytes_per_dispatch"]["with_gfx950_x2_correction_upper_bound"])
                        traffic = t_read + t_write
                    except Exception:
                        traffic = None
                    vi = d.get("SQ_INSTS_VALU_per_sample")
                    if vi and k_ms:
                        rate = vi * local_samples / (k_ms * 1e-3)
                        issue = {"bound": "valu_issue", "valu_wave_instr_per_sample": round(vi, 1), "cycles_per_wave_instr": 2,
                                 "peak_wave_instr_per_s": VALU_WAVE_INSTR_PER_S, "achieved_wave_instr_per_s": round(rate, 1),
                                 "frac": round(rate / VALU_WAVE_INSTR_PER_S, 4), "lane_fill": round(d.get("valu_lane_utilisation", 0.0), 4),
                                 "note": "instruction count per sample from the committed PMC pass (a property of the build), rate from THIS run's "
                                         "kernel time; frac x lane_fill = share of the 78.6 T lane-instr/s the kernel's useful lanes occupy"}
            traffic_note = (f"HBM bytes per launch from the committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command ({prof_path}), "
                            "not re-measured in this run")
            if live_traffic and not spp_override and bvh == "reference" and not fast_accel and not handed_over_tree and not empirical:
                live, why = measure_counters_live(name)
                if live is not None:
                    traffic_note = ("HBM bytes of one launch MEASURED IN THIS RUN: one-step child runs of this script under rocprofv3 --pmc "
                                    f"(FETCH_SIZE x 2 as the gfx950 upper bound: {live['read_bytes_upper_bound']:.0f} B read, WRITE_SIZE: "
                                    f"{live['written_bytes']:.0f} B written); committed profile ({prof_path}): {traffic}")
                    t_read, t_write = live["read_bytes_upper_bound"], live["written_bytes"]
                    traffic = t_read + t_write
                    # What the HBM writes are made of, from the L2's request counters: the kernel's vector-memory writes are the 64-bit
                    # atomic adds of the fixed-point pixel sums (a unit's flush of its tile sums, and the samples that finish after their
                    # wave has moved on to the next unit) and plain stores (register spills, the redo queue entries of exact re-treeing).
                    # Every atomic ends as a write request to memory (profiles/r04/c2_pmc_summary.json: TCC_EA0_WRREQ = TCC_ATOMIC +
                    # TCC_WRITE to 0.1 %), so the plain stores' share of the requests bounds the spills' share of the written bytes.
                    plain, atomics = live["TCC_WRITE_sum"], live["TCC_ATOMIC_sum"]
                    scratch_share = round(plain / max(1.0, plain + atomics), 4)
                    scratch_detail = {"l2_plain_store_requests": round(plain), "l2_atomic_requests": round(atomics),
                                      "written_bytes_per_request": round(t_write / max(1.0, plain + atomics), 1),
                                      "note": "plain stores = register spills + redo queue entries (8 B per requeued sample); atomics = tile-sum "
                                              "flushes + stragglers' direct adds"}
                    if k_ms:
                        vi = live["SQ_INSTS_VALU"] / local_samples
                        rate = live["SQ_INSTS_VALU"] / (k_ms * 1e-3)
                        issue = {"bound": "valu_issue", "valu_wave_instr_per_sample": round(vi, 1), "salu_wave_instr_per_sample": round(live["SQ_INSTS_SALU"] / local_samples, 1),
                                 "cycles_per_wave_instr": 2, "peak_wave_instr_per_s": VALU_WAVE_INSTR_PER_S, "achieved_wave_instr_per_s": round(rate, 1),
                                 "frac": round(rate / VALU_WAVE_INSTR_PER_S, 4),
                                 "lane_fill": round(live["SQ_THREAD_CYCLES_VALU"] / (64.0 * live["SQ_ACTIVE_INST_VALU"]), 4),
                                 "note": "instruction counts of one launch MEASURED IN THIS RUN (rocprofv3 --pmc child run), rate from this run's kernel "
                                         "time; frac x lane_fill = share of the 78.6 T lane-instr/s the kernel's useful lanes occupy"}
                else:
                    traffic_note += f" (a live measurement was tried and failed: {why})"
            roof = None
            if k_ms:
                achieved = bps * local_samples / (k_ms * 1e-3) / 1e9
                roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                        "traffic_read": t_read, "traffic_write": t_write, "scratch_write_share": scratch_share, "scratch_detail": scratch_detail,
                        "traffic_note": traffic_note,
                        "algorithmic_bytes_per_sample": round(bps, 1), "kernel_ms": round(k_ms, 3),
                        "algorithmic_bytes_per_sample_walked": round(bps_walked, 1),
                        "frac_walked": round(achieved * bps_walked / bps / HBM_PEAK_GBPS, 4),
                        "walked": walked,
                        "walked_note": "the bytes of the walk the device PERFORMS: on the tree as handed over the oracle's visits without the "
                                       "whole-tree walks of NaN / infinite rays (the device skips them in sphere-only scenes: their outcome is a "
                                       "miss); on a rebuilt tree the visits of that tree, second walks included (`walked`), counted by the "
                                       "kernel's per-lane code on the host.  algorithmic_bytes_per_sample (SURVEY 8d: the oracle on the tree "
                                       "handed over) stays the contract value.",
                        "note": "algorithmic bytes (SURVEY 8d record sizes x oracle visit counts); the scene is LDS/L2 resident, so this is not "
                                "a physical bound (it can exceed 1): the physical one is `issue`",
                        "issue": issue}
            physical = physical_bounds(c, local_samples, k_ms * 1e-3, bool(info.lds_bytes), issue, walked) if k_ms else None
            res = {"value": round(value, 2), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup,
                   "physical": physical, "kernel_ms": k_ms, "requeued_samples": ds.last_requeued_samples(),
                   "label": label, "integrator": "scatter" if hs.integrator else "pdf", "bvh_items": info.n_items, "tree": TREE_NAMES.get(info.tree),
                   "scene_in_lds": bool(info.lds_bytes), "verified": verified, "roofline": roof, "cpu_baseline": cpu,
                   "tree_code": info.tree, "_frame": final}
        if args.in_library and rank == 0:
            # The same self-diagnosing block for ONE process driving N devices (vk_scene_create_multi): what ran where — every share's
            # device, PCI bus id, whether it can address devices[0] (else its slab goes through host memory), its kernel time in the last
            # timed frame — how the slabs travelled, and the gathered frame against ONE device's render of the whole frame.
            from vecchio_amd import ffi
            parts = ds.parts()
            ds1 = DeviceScene(hs.desc, device=in_lib_devices[0])
            ref, _ = ds1.render(cam, hs.params(width, spp, depth, seed=2))
            ds1.close()
            same = bool(np.array_equal(final.view(np.uint32), ref.view(np.uint32)))
            res["in_library"] = {"devices": in_lib_devices, "distinct_devices": len(set(in_lib_devices)), "parts": parts,
                                 "gather": {ffi.VK_GATHER_NONE: "none (one device)", ffi.VK_GATHER_PEER_COPY: "hipMemcpyPeerAsync",
                                            ffi.VK_GATHER_RCCL: "RCCL (grouped ncclSend / ncclRecv)"}.get(info.gather),
                                 "gather_backends": {"peer_copy": True, "rccl_loadable": bool(ffi.load_device_lib().vk_gather_backends() & 2)},
                                 "slowest_part_kernel_ms": max(p["kernel_ms"] for p in parts),
                                 "gathered_image_equals_one_gpu_render": same}
            print(f"bench.py: in-library, {len(parts)} parts on {len(set(in_lib_devices))} device(s), gather by {res['in_library']['gather']}, "
                  f"gathered image bit-identical to the 1-GPU render: {same}", file=sys.stderr)
            assert same, "the in-library multi-device image differs from the one-GPU render"
        if world > 1:
            # what a driver needs to see that N ranks really ran: every rank's device, and the gathered image against ONE device's
            # render of the whole frame (bit-identical by construction: order-independent fixed-point pixel sums)
            names = [None] * world
            me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device": torch.cuda.get_device_name(dev_index),
                  "pci_bus_id": getattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id", None), "pid": os.getpid()}
            dist.all_gather_object(names, me)
            if rank == 0:
                ds1 = DeviceScene(hs.desc, device=dev_index)
                ref, _ = ds1.render(cam, hs.params(width, spp, depth, seed=2))
                if args.rgb8:
                    # the gathered frame is bytes (to_color fused into the exchange): against ONE device's own RGB8 output of the whole
                    # frame; the oracle check then runs on that device's f32 frame
                    from vecchio_amd import ffi
                    ref8, _ = ds1.render(cam, hs.params(width, spp, depth, seed=2, output_format=ffi.VK_OUTPUT_RGB8))
                    same = bool((full.cpu().numpy() == ref8).all())
                    if not args.no_verify:
                        res["verified"] = verify_against_oracle(O, hs, cam, ref, width, spp, depth)
                        res["verified"]["note"] = "the one-GPU f32 frame, whose RGB8 output the gathered frame equals byte for byte"
                else:
                    same = bool((full.cpu().numpy() == ref).all())
                ds1.close()
                res["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": names,
                                      "distinct_devices": len({(n["device_index"]) for n in names}),
                                      "gathered_image_equals_one_gpu_render": same,
                                      "self_launched": os.environ.get("VK_BENCH_SELF_LAUNCHED") == "1"}
                print(f"bench.py: {world} ranks ({dist.get_backend()}), gathered image bit-identical to the 1-GPU render: {same}", file=sys.stderr)
                assert same, "the gathered multi-rank image differs from the one-GPU render"
        ds.close()
        hs.close()
        return res

    main_res = run_workload(args.workload, args.steps, args.warmup, args.spp, want_cpu=(n_gpus == 1 and not args.no_cpu), bvh=args.bvh,
                            fast_accel=args.fast_accel, live_traffic=(n_gpus == 1 and world == 1 and not args.no_traffic and not rehearsal),
                            handed_over_tree=args.reference_tree, empirical=args.empirical_trees)
    also = []
    handed_over = None
    frames = {}            # workload -> its frame on the tree as handed over
    exact_mismatch = []    # a tree that claims the handed-over tree's results and does not deliver them
    if n_gpus == 1 and not args.no_also and not args.spp and not args.reference_tree and rank == 0:
        # the same workload on the tree of the description alone: the number the rebuilt tree has to be read against
        r = run_workload(args.workload, 1, 1, 0, handed_over_tree=True)
        # ... and the frame the default has to EQUAL, bit for bit: the last timed frame of the headline against this one (exact re-treeing
        # claims BVHNode::hit's results on the tree as handed over, accel.rs:58-83; pixel sums are order independent)
        frames[args.workload] = r.pop("_frame")
        same = bool(np.array_equal(main_res["_frame"].view(np.uint32), frames[args.workload].view(np.uint32)))
        handed_over = {"value": r["value"], "unit": "Msamples/s", "ms_per_step": r["ms_per_step"], "kernel_ms": r["kernel_ms"], "steps": 1, "warmup": 1,
                       "workload": r["label"], "verified": r["verified"], "frame_bit_identical": same,
                       "frame_bit_identical_note": "the headline's last timed frame (tree: %s) == this frame, all %d x %d x 3 floats" % (
                           main_res["tree"], frames[args.workload].shape[1], frames[args.workload].shape[0])}
        if not same:
            exact_mismatch.append(f"{args.workload} default ({main_res['tree']}) differs from the tree as handed over in "
                                  f"{int((main_res['_frame'] != frames[args.workload]).any(axis=2).sum())} pixels")
    if n_gpus == 1 and not args.no_also and args.workload == "C2" and not args.spp:
        # (C5: on the tree as handed over first — the frame its default, the near form of exact re-treeing, must equal)
        for name, spp_o, fa, ho, emp in (("C4", 0, False, False, False), ("C3", 0, False, False, False), ("C5", 0, False, True, False),
                                         ("C5", 0, False, False, False), ("C5", 0, False, False, True), ("C2", 0, False, False, True),
                                         ("C2", 0, True, False, False)):
            r = run_workload(name, 1, 0, spp_o, fast_accel=fa, handed_over_tree=ho, empirical=emp)
            frame = r.pop("_frame")
            ident = None
            if r["tree_code"] == 0:
                frames.setdefault(name, frame)                # (walked as handed over: C5's default, C3, C4)
            elif name in frames:
                ident = bool(np.array_equal(frame.view(np.uint32), frames[name].view(np.uint32)))
                if not ident and r["tree_code"] in (1, 4, 5):
                    exact_mismatch.append(f"{r['label']} ({r['tree']}) differs from the tree as handed over")
            del frame
            also.append({"workload": r["label"], "tree": r["tree"], "Msamples_per_s": r["value"], "ms_per_step": r["ms_per_step"], "steps": 1,
                         # rebuilt trees only: this frame == the same workload's frame on the tree as handed over (proven trees must; the
                         # empirical form and VK_SCENE_FAST_ACCEL are measured)
                         "frame_bit_identical_to_handed_over": ident,
                         # one step, no warm-up: `Msamples_per_s` includes the variant's first-use costs (code-object load, buffer
                         # allocation); the kernel-time rate does not and is the one comparable with the headline
                         "requeued_samples": r["requeued_samples"],
                         "kernel_ms": r["kernel_ms"], "kernel_Msamples_per_s": round(WORKLOAD_SAMPLES(name) / r["kernel_ms"] / 1e3, 2) if r["kernel_ms"] else None,
                         "physical": r["physical"],
                         "verified": r["verified"], "roofline_frac": r["roofline"]["frac"] if r["roofline"] else None,
                         "issue_frac": (r["roofline"] or {}).get("issue", None) and r["roofline"]["issue"]["frac"]})
    if rank == 0:
        r = main_res
        r.pop("_frame", None)
        mode = "one process per GPU" if not args.in_library else "one process, vk_scene_create_multi (in-library tile deal + gather)"
        out = {
            "metric": "Msamples/sec (pixels x spp)", "value": r["value"], "unit": "Msamples/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {"workload": r["label"], "scene_seed": 1, "render_seed": 2,
                       "integrator": r["integrator"], "tiles": "8x8 round-robin over ranks", "multi_gpu": mode,
                       "bvh_builder": args.bvh, "bvh_items": r["bvh_items"], "scene_in_lds": r["scene_in_lds"],
                       # scenes of spheres only may be walked on a tree rebuilt over the reference's leaf units; where the rebuilt walk's
                       # winner is not certain to be the reference's, the tree handed over decides (vk_trace.h segment_unsafe).
                       # `requeued_samples`: the samples of the last step that went through the second launch for that (scenes in LDS)
                       "tree": r["tree"],
                       "requeued_samples": r["requeued_samples"], "also": also},
            "verified": r["verified"], "roofline": r["roofline"], "physical": r["physical"], "cpu_baseline": r["cpu_baseline"],
            "handed_over_tree": handed_over,
        }
        if r.get("distributed"):
            out["distributed"] = r["distributed"]
        if r.get("in_library"):
            out["in_library"] = r["in_library"]
        print(json.dumps(out), flush=True)
        if r["verified"] and not r["verified"].get("ok", False):
            print("bench.py: the timed framebuffer does NOT match the oracle", file=sys.stderr)
            sys.exit(3)
        if exact_mismatch:
            print("bench.py: " + "; ".join(exact_mismatch), file=sys.stderr)
            sys.exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
