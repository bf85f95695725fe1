"""GPU-box helper: one viewpoint of a sphere world, default flags against VK_SCENE_REFERENCE_TREE, differing pixels.
    python tools/experiments/view_check.py stress_spheres:60 900 15 0  0 0 0"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from descs import camera
from vecchio_amd import DeviceScene, HostScene, ffi
scene = sys.argv[1]; lf = tuple(float(x) for x in sys.argv[2:5]); la = tuple(float(x) for x in sys.argv[5:8])
imgs = {}
for flags in (0, ffi.VK_SCENE_REFERENCE_TREE):
    hs = HostScene(scene, 1); hs.desc.contents.flags = flags
    cam = camera(lf, la, vfov=35.0, aspect=16.0 / 9.0, aperture=0.0, focus=10.0)
    ds = DeviceScene(hs.desc)
    imgs[flags] = ds.render(cam, hs.params(384, 24, 50, seed=6))[0]; tree = ds.info().tree
    ds.close(); hs.close()
a, b = imgs[0], imgs[ffi.VK_SCENE_REFERENCE_TREE]
print(scene, lf, "differing pixels:", int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum()), "lib", os.environ.get("VK_DEVICE_LIB", "default"))
