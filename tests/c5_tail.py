import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vecchio_amd import DeviceScene, HostScene
hs = HostScene("stress_spheres:500", 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc)
for spp in (1, 2, 4, 8, 16, 32):
    p = hs.params(4096, spp, 50)
    ds.render(cam, p)
    ms = min(ds.render(cam, p)[1].kernel_ms for _ in range(2))
    print(f"spp {spp}: {ms:.1f} ms  ({4096*4096*spp/ms/1e3:.1f} Msamples/s)", flush=True)
for depth in (1, 2, 5, 10, 50):
    p = hs.params(4096, 4, depth)
    ms = min(ds.render(cam, p)[1].kernel_ms for _ in range(2))
    print(f"depth {depth} spp 4: {ms:.1f} ms", flush=True)
