"""GPU-box helper (not a test): kernel-time throughput of a scene under several vk_scene_desc.flags in one process.
    python tools/experiments/ab_flags.py final_scene:800:256 cornell_box:1024:256 [--flags 0,1]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from vecchio_amd import HostScene, DeviceScene, ffi
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [0, ffi.VK_SCENE_FAST_ACCEL]
for a in sys.argv[1:]:
    if a.startswith("--flags="): flags = [int(x) for x in a[8:].split(",")]
for job in args:
    parts = job.split(":"); name, w, spp = ":".join(parts[:-2]), int(parts[-2]), int(parts[-1])
    for fl in flags:
        hs = HostScene(name, 1); hs.desc.contents.flags = fl
        cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(w, spp, 50)
        ds.render(cam, p); best = 0.0
        for _ in range(2):
            _, st = ds.render(cam, p); best = max(best, st.samples / st.kernel_ms / 1e3)
        print(f"{name} flags={fl}: {best:.1f} Msamples/s  tree={ds.info().tree} items={ds.info().n_items if hasattr(ds.info(),'n_items') else '-'}", flush=True)
        ds.close(); hs.close()
