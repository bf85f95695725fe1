"""GPU-box helper (not a test): kernel-time throughput of one scene with and without environment settings that the library reads
at vk_scene_create, in one process and interleaved.
    python tools/experiments/ab_env.py random_spheres_iow:1920:256 VK_ORDER_POINT=0,100,0 [NAME=value ...]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from vecchio_amd import HostScene, DeviceScene
job = sys.argv[1].split(":"); name, w, spp = ":".join(job[:-2]), int(job[-2]), int(job[-1])
settings = [None] + sys.argv[2:]
res = {}
for rep in range(3):
    for st in settings:
        if st: k, v = st.split("=", 1); os.environ[k] = v
        hs = HostScene(name, 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(w, spp, 50)
        if st: del os.environ[k]
        ds.render(cam, p)
        _, s = ds.render(cam, p)
        res.setdefault(st or "default", []).append(s.samples / s.kernel_ms / 1e3)
        ds.close(); hs.close()
for k, v in res.items():
    print(f"{name} {k}: " + " ".join(f"{x:.1f}" for x in v) + f"  best {max(v):.1f} Msamples/s", flush=True)
