"""How often does VK_SCENE_FAST_ACCEL change a sample?  Pixel sums are order independent, so the image of the re-treed scene is
bit-identical to the exact one unless some SAMPLE took a different path.  Usage (GPU box): python tools/experiments/fast_accel_diff.py [scene width spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene, ffi  # noqa: E402

jobs = [("random_spheres_iow", 1920, 1024), ("random_spheres_demo", 960, 256), ("stress_spheres:500", 2048, 16)]
if len(sys.argv) == 4:
    jobs = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))]
for name, w, spp in jobs:
    imgs = []
    for flags in (0, ffi.VK_SCENE_FAST_ACCEL):
        hs = HostScene(name, 1)
        hs.desc.contents.flags = flags
        cam = hs.next_camera()
        p = hs.params(w, spp, 50)
        ds = DeviceScene(hs.desc)
        ds.render(cam, p)
        img, st = ds.render(cam, p)
        imgs.append((img, st.kernel_ms, ds.info().n_items))
        ds.close(); hs.close()
    (a, ta, na), (b, tb, nb) = imgs
    diff = (a != b).any(axis=2)
    print(f"{name} {w}x{a.shape[0]}x{spp}: exact {ta:.1f} ms ({na} items), fast accel {tb:.1f} ms ({nb} items); "
          f"{int(diff.sum())} of {diff.size} pixels differ ({a.shape[0] * w * spp} samples), max |d| {float(np.abs(a - b).max()):.3g}", flush=True)
