import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene, ffi
name, w, spp = "random_spheres_iow", 1920, 256
def run(flags, n=2):
    hs = HostScene(name, 1); hs.desc.contents.flags = flags
    cam = hs.next_camera(); p = hs.params(w, spp, 50); ds = DeviceScene(hs.desc)
    out = [ds.render(cam, p)[0].copy() for _ in range(n)]
    ds.close(); hs.close(); return out
r = run(ffi.VK_SCENE_REFERENCE_TREE, 1)[0]
e = run(0, 3)
for i, x in enumerate(e):
    d = np.argwhere((x != r).any(axis=2))
    print("exact run", i, "vs reference:", len(d), "pixels differ")
    for (y, xx) in d[:6]:
        print("   ", y, xx, r[y, xx], x[y, xx], (x[y, xx].astype(np.float64) - r[y, xx]) * spp * 2**26)
print("exact runs equal to each other:", [bool((e[0] == x).all()) for x in e[1:]])
