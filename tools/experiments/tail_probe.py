"""Frame time against spp for each BASELINE scene: T = a + b*spp.  A large intercept `a` is a serial tail (one lane working alone
at the end of the launch) — how the NaN-ray walks of the 1 M-sphere scene were found.  Usage (GPU box): python tools/experiments/tail_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

for name, w, spps in (("random_spheres_iow", 1920, (8, 16, 32, 64, 128)), ("cornell_box", 1024, (16, 32, 64, 128, 256)),
                      ("final_scene", 800, (16, 32, 64, 128, 256)), ("stress_spheres:500", 4096, (1, 2, 4, 8, 16))):
    hs = HostScene(name, 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc)
    ts = []
    for spp in spps:
        p = hs.params(w, spp, 50)
        ds.render(cam, p)
        ts.append(min(ds.render(cam, p)[1].kernel_ms for _ in range(3)))
    b, a = np.polyfit(np.array(spps, float), np.array(ts), 1)
    n = w * p.height
    print(f"{name}: " + "  ".join(f"{s} spp {t:.1f} ms" for s, t in zip(spps, ts)) + f"   fit: {a:.2f} ms + {b:.3f} ms/spp = {n / b / 1e3:.0f} Msamples/s asymptotically", flush=True)
    ds.close(); hs.close()
