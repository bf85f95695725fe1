"""GPU-box helper: throughput of a stress world seen from INSIDE the layer (every segment starts near the spheres), with whatever
environment is set (VK_GRID_GLOBAL=1: the grid form from global memory).  python tools/experiments/grid_inside_view.py stress_spheres:100"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from descs import camera
from vecchio_amd import DeviceScene, HostScene
for scene in sys.argv[1:]:
    for lf, la, tag in (((3.0, 0.6, 2.0), (10.0, 0.3, 9.0), "inside"), ((3.0, 6.0, 2.0), (14.0, 0.0, 11.0), "low above")):
        hs = HostScene(scene, 1)
        cam = camera(lf, la, vfov=50.0, aspect=16.0 / 9.0, aperture=0.0, focus=10.0)
        ds = DeviceScene(hs.desc); p = hs.params(1024, 64, 50, seed=6)
        ds.render(cam, p); best = 0.0
        for _ in range(2):
            _, st = ds.render(cam, p); best = max(best, st.samples / st.kernel_ms / 1e3)
        print(f"{scene} {tag}: {best:.1f} Msamples/s tree={ds.info().tree} requeued={ds.last_requeued_samples()} of {st.samples}", flush=True)
        ds.close(); hs.close()
