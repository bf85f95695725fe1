"""Exact re-treeing against the handed-over tree over several worlds (scene seeds) and render seeds: every image must be bit-identical.
The default form (round 5: the near form, staged in LDS for the InOneWeekend worlds) and what VK_SCENE_EMPIRICAL_TREES adds
(VK_SCENE_EMPIRICAL_TREES: the stress worlds, whose default is the tree as handed over).
Usage (GPU box): python tools/experiments/exact_retree_seeds.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene, ffi  # noqa: E402

total = 0
bad = 0
for name, w, spp, seeds in (("random_spheres_iow", 1920, 128, range(2, 82)), ("stress_spheres:500", 4096, 4, range(2, 6)),
                            ("stress_spheres:200", 2048, 8, range(2, 14)), ("stress_spheres:100", 2048, 8, range(2, 10)),
                            ("stress_spheres:60", 1024, 32, range(2, 8))):
    for seed in seeds:
        imgs = []
        for flags in (ffi.VK_SCENE_REFERENCE_TREE, 0 if name == "random_spheres_iow" else ffi.VK_SCENE_EMPIRICAL_TREES):
            hs = HostScene(name, seed)
            hs.desc.contents.flags = flags
            cam = hs.next_camera()
            p = hs.params(w, spp, 50, seed=seed * 7 + 1)
            ds = DeviceScene(hs.desc)
            img, st = ds.render(cam, p)
            imgs.append(img)
            rq = ds.last_requeued_samples()
            tree = ds.info().tree
            ds.close(); hs.close()
        d = int((imgs[0] != imgs[1]).any(axis=2).sum())
        total += st.samples
        bad += d
        print(f"{name} scene seed {seed}: {w}x{imgs[0].shape[0]}x{spp} = {st.samples} samples, {rq} requeued, tree form {tree}, {d} pixels differ", flush=True)
print(f"total {total} samples, {bad} differing pixels", flush=True)
