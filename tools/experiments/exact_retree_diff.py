"""Exact re-treeing against the tree as handed over, on the GPU: the two images must be bit-identical (pixel sums are order
independent, so they are unless some SAMPLE differs).  Usage (GPU box): python tools/experiments/exact_retree_diff.py [scene width spp ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene, ffi  # noqa: E402

jobs = [("random_spheres_iow", 1920, 256), ("random_spheres_demo", 960, 64), ("stress_spheres:500", 2048, 16)]
a = sys.argv[1:]
if a:
    jobs = [(a[i], int(a[i + 1]), int(a[i + 2])) for i in range(0, len(a), 3)]
lib = ffi.load_device_lib()
for name, w, spp in jobs:
    imgs = []
    # (the stress worlds are rebuilt only on request: their default is the tree as handed over)
    for flags in (ffi.VK_SCENE_REFERENCE_TREE, ffi.VK_SCENE_EMPIRICAL_TREES if name.startswith("stress") else 0):
        hs = HostScene(name, 1)
        hs.desc.contents.flags = flags
        cam = hs.next_camera()
        p = hs.params(w, spp, 50)
        ds = DeviceScene(hs.desc)
        ds.render(cam, p)
        best = None
        for _ in range(2):
            img, st = ds.render(cam, p)
            best = st.kernel_ms if best is None else min(best, st.kernel_ms)
        rq = C.c_uint64(0)
        rc = lib.vk_scene_last_requeued_samples(ds._h, C.byref(rq))
        imgs.append((img, best, ds.info().n_items, rq.value, rc, st.samples))
        ds.close(); hs.close()
    (x, tx, nx, _, _, ns), (y, ty, ny, rq, rc, _) = imgs
    diff = (x != y).any(axis=2)
    print(f"{name} {w}x{x.shape[0]}x{spp}: handed-over tree {tx:.1f} ms = {ns / tx / 1e3:.0f} Msamples/s ({nx} items); exact re-tree {ty:.1f} ms = "
          f"{ns / ty / 1e3:.0f} Msamples/s ({ny} items), {rq} samples requeued ({rq / ns:.2%}, rc {rc}); {int(diff.sum())} of {diff.size} pixels differ, "
          f"max |d| {float(np.abs(x - y).max()):.3g}", flush=True)
