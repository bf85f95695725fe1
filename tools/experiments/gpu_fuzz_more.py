"""Extra randomised GPU parity runs beyond the test suite (GPU box): python tools/experiments/gpu_fuzz_more.py [n_graphs n_crowds n_sphere_worlds]
Every scene: HIP path through the C ABI vs the oracle per sample (equal draw counts, |dRGB| < 1e-4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_ffi as O  # noqa: E402
from test_fuzz_scenes import Gen  # noqa: E402
from test_gpu_parity import compare_samples, device_samples  # noqa: E402
from test_retree import Crowd, SphereCrowd  # noqa: E402
from vecchio_amd import DeviceScene  # noqa: E402

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_crowds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_spheres = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # worlds of spheres only: the default path there is exact re-treeing
bad = 0
for kind, n, base, cls in (("graph", n_graphs, 7000, Gen), ("crowd", n_crowds, 8000, Crowd), ("spheres", n_spheres, 9000, SphereCrowd)):
    for seed in range(base, base + n):
        desc, cam, p = (cls(seed, nasty=seed % 2 == 1) if cls is SphereCrowd else cls(seed)).build()
        ds = DeviceScene(desc)
        img_d, ps_d = device_samples(ds, cam, p)
        img_o, ps_o = O.render_samples(desc, cam, p)
        try:
            compare_samples(ps_o, ps_d, img_o, img_d)
        except AssertionError as e:
            bad += 1
            print(f"{kind} seed {seed}: {str(e)[:120]}", flush=True)
        ds.close()
    print(f"{kind}: {n} scenes done, {bad} failures so far", flush=True)
sys.exit(1 if bad else 0)
