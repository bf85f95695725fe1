"""One rank's 1/8 share of C2 at a few sample counts (GPU box; a profiling target: python tools/experiments/one_share.py [spp ...])"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

hs = HostScene("random_spheres_iow", 1)
cam = hs.next_camera()
ds = DeviceScene(hs.desc)
for spp in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    p = hs.params(1920, spp, 50, seed=2, tile_rank=0, tile_world=8)
    ds.render(cam, p)
    print(spp, "spp:", " ".join(f"{ds.render(cam, p)[1].kernel_ms:.2f}" for _ in range(3)), "ms", flush=True)
