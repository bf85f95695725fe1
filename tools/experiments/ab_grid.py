import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from vecchio_amd import HostScene, DeviceScene, ffi
for job in sys.argv[1:]:
    parts = job.split(":"); name, w, spp = ":".join(parts[:-2]), int(parts[-2]), int(parts[-1])
    hs = HostScene(name, 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(w, spp, 50)
    ds.render(cam, p); best = 0.0
    for _ in range(2):
        _, st = ds.render(cam, p); best = max(best, st.samples / st.kernel_ms / 1e3)
    print(f"{name}: {best:.1f} Msamples/s tree={ds.info().tree} lds={st.scene_in_lds} requeued={ds.last_requeued_samples()} of {st.samples}", flush=True)
    ds.close(); hs.close()
