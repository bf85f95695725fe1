import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from vecchio_amd import HostScene, DeviceScene
for seed in range(1, 9):
    hs = HostScene("final_scene", seed); d = hs.desc.contents
    dup7 = sum(1 for i in range(d.n_bvh) if d.bvh[i].left == d.bvh[i].right and ((d.bvh[i].left >> 28) & 0xF) == 7)
    cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(800, 512, 50)
    ds.render(cam, p); best = 0
    for _ in range(2):
        _, st = ds.render(cam, p); best = max(best, st.samples / st.kernel_ms / 1e3)
    print(f"final_scene world seed {seed}: cluster in a len-1 node: {bool(dup7)}; {best:.1f} Msamples/s", flush=True)
    ds.close(); hs.close()
