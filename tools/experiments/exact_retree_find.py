import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene, ffi
name, w, spp, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
imgs = []
for flags in (ffi.VK_SCENE_REFERENCE_TREE, 0):
    hs = HostScene(name, seed); hs.desc.contents.flags = flags; cam = hs.next_camera(); p = hs.params(w, spp, 50, seed=seed * 7 + 1)
    ds = DeviceScene(hs.desc); img, st = ds.render(cam, p); imgs.append(img); ds.close(); hs.close()
for (y, x) in np.argwhere((imgs[0] != imgs[1]).any(axis=2)):
    print("PIXEL", y, x, imgs[0][y, x], imgs[1][y, x], (imgs[1][y, x].astype(np.float64) - imgs[0][y, x]) * spp)
