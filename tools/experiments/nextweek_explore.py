"""GPU-box helper (not a test): renders final_scene_nextweek 900x900 with the HIP path and leaves the linear f32
image under gpurun_out/ so that region rectangles for tests/golden/make_nextweek_regions.py can be chosen against
sample/thenextweek.png in the build container.   python tools/experiments/nextweek_explore.py [spp] [depth]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vecchio_amd import DeviceScene, HostScene

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for name in ("final_scene_nextweek", "final_scene"):
    hs = HostScene(name, 1)
    cam = hs.next_camera()
    ds = DeviceScene(hs.desc)
    p = hs.params(900, spp, depth)
    img, st = ds.render(cam, p)
    print(name, p.width, p.height, spp, f"{st.kernel_ms:.0f} ms", "finite", bool(np.isfinite(img).all()), "mean", img.reshape(-1, 3).mean(0))
    np.save(f"gpurun_out/{name}_900_{spp}.npy", img.astype(np.float16) if False else img)
    ds.close()
