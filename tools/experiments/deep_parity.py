"""Per-sample parity at a scale the test suite cannot afford on every run (GPU box): the HIP path through the C ABI against the
oracle, EVERY sample of a frame (equal draw counts = same path, same finite flag, relative |dRGB| < 2e-5), in slabs of rows so that
the per-sample dumps fit in memory.  Usage: python tools/experiments/deep_parity.py [scene width spp depth] — default C2's frame at 32 spp
(66 M samples; rare events down to ~1e-7 per sample are then seen), then the Cornell box, the final scene and the 1 M-sphere scene."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_ffi as O  # noqa: E402
from test_gpu_parity import compare_samples, device_samples  # noqa: E402
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

jobs = [("random_spheres_iow", 1920, 32, 50), ("cornell_box", 1024, 32, 50), ("final_scene", 800, 32, 50), ("stress_spheres:500", 2048, 4, 50),
        ("stress_spheres:500+empirical", 2048, 4, 50)]
if len(sys.argv) == 5:
    jobs = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))]
for name, width, spp, depth in jobs:
    empirical = name.endswith("+empirical")
    hs = HostScene(name.replace("+empirical", ""), 1)
    if empirical:
        from vecchio_amd import ffi
        hs.desc.contents.flags = ffi.VK_SCENE_EMPIRICAL_TREES
    cam = hs.next_camera()
    ds = DeviceScene(hs.desc)
    p = hs.params(width, spp, depth, seed=2)
    t0 = time.time()
    img_d, ps_d = device_samples(ds, cam, p)
    t1 = time.time()
    img_o, ps_o = O.render_samples(hs.desc, cam, p)
    t2 = time.time()
    n = ps_d.shape[0]
    try:
        compare_samples(ps_o, ps_d, img_o, img_d)
        dropped = int((~np.isfinite(ps_o[:, :3]).all(1)).sum())
        print(f"{name} {width}x{p.height}x{spp} depth {depth}: {n} samples, every one on the oracle's path (equal draw counts), {dropped} non-finite "
              f"samples dropped by both, max |d pixel| {np.abs(img_o - img_d).max():.2e}  (device {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s)", flush=True)
    except AssertionError as e:
        print(f"{name}: MISMATCH {str(e)[:200]}", flush=True)
        sys.exit(1)
    ds.close(); hs.close()
