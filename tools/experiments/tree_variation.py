"""GPU-box helper: how much of a frame depends on WHICH tree BVHNode::new built?  The reference's tree is random (split axes
from thread_rng, accel.rs:99-100), and an f32 Sphere::hit that reports a hit a hair outside the sphere's own bounding box is found or
not depending on the enclosing boxes of the tree at hand.  Renders the same world three ways — the reference-style tree of build
stream A, the reference-style tree of build stream B (same objects: `+treeseed`), and tree A with VK_SCENE_FAST_ACCEL (library
SAH rebuild) — and counts the pixels whose fixed-point sums differ (order-independent sums: any sample that took another path shows).
    python tools/experiments/tree_variation.py [scene=stress_spheres:500] [width=2048] [spp=8]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vecchio_amd import DeviceScene, HostScene, ffi

scene = sys.argv[1] if len(sys.argv) > 1 else "stress_spheres:500"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
imgs = {}
for label, name, flags in (("tree A", scene, 0), ("tree B", scene + "+treeseed:7", 0), ("tree C", scene + "+treeseed:8", 0), ("A + FAST_ACCEL", scene, ffi.VK_SCENE_FAST_ACCEL)):
    hs = HostScene(name, 1)
    hs.desc.contents.flags = flags
    cam = hs.next_camera()
    ds = DeviceScene(hs.desc)
    p = hs.params(width, spp, 50)
    img, st = ds.render(cam, p)
    imgs[label] = img
    print(f"{label:16s} {st.samples / st.kernel_ms / 1e3:8.1f} Msamples/s  items {ds.info().n_items}", flush=True)
    ds.close(); hs.close()
n = width * imgs["tree A"].shape[0]
keys = list(imgs)
for i in range(len(keys)):
    for j in range(i + 1, len(keys)):
        a, b = imgs[keys[i]], imgs[keys[j]]
        diff = (a.view(np.uint32) != b.view(np.uint32)).any(2)
        big = np.abs(a - b).max(2) > 1e-4
        print(f"{keys[i]:16s} vs {keys[j]:16s}: {int(diff.sum()):8d} of {n} pixels differ ({diff.mean() * 100:.4f} %), {int(big.sum())} by more than 1e-4; "
              f"<= {diff.mean() / spp * 100:.5f} % of samples", flush=True)
