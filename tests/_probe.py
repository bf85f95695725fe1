import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vecchio_amd import HostScene, DeviceScene
hs = HostScene("stress_spheres:500", 1); cam = hs.next_camera()
ds = DeviceScene(hs.desc)
for w, spp in ((1024, 128),):
    p = hs.params(w, spp, 50)
    img, st = ds.render(cam, p)
    print(f"waves={os.environ.get('VK_GLOBAL_WAVES')} {w}x{p.height}x{spp}: {st.samples/st.kernel_ms/1e3:.2f} Msamples/s ({st.kernel_ms:.0f} ms)", flush=True)
