"""Kernel time of one rank's 1/N share of C2 against spp: T = a + b * spp; `a` is what a launch costs whatever its length (ramp-up, the last
units, the deepest paths, the second launch of exact re-treeing).  Usage (GPU box): python tools/experiments/share_intercept.py [N] [max_depth]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 50
hs = HostScene("random_spheres_iow", 1)
cam = hs.next_camera()
ds = DeviceScene(hs.desc)
spps, ts = (256, 512, 1024, 2048, 4096), []
for spp in spps:
    p = hs.params(1920, spp, depth, seed=2, tile_rank=0, tile_world=n)
    ds.render(cam, p)
    ts.append(min(ds.render(cam, p)[1].kernel_ms for _ in range(3)))
b, a = np.polyfit(np.array(spps, float), np.array(ts), 1)
print(f"1/{n} share, depth {depth}: " + "  ".join(f"{s} spp {t:.2f} ms" for s, t in zip(spps, ts)) + f"   fit: {a:.2f} ms + {b * 1024:.2f} ms per 1024 spp", flush=True)
