"""Is the fixed cost of a partition launch real or an artefact of idling between synchronous calls?  Renders one rank's 1/N share
K times back to back on one stream (no host synchronisation in between) and prints the time per launch next to the synchronous
figure of partition_probe.py.  Usage (GPU box): python tools/experiments/backtoback_probe.py [N] [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from vecchio_amd import DeviceScene, HostScene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.cuda.init()
hs = HostScene("random_spheres_iow", 1)
cam = hs.next_camera()
ds = DeviceScene(hs.desc)
for world in (1, n):
    p = hs.params(1920, 1024, 50, seed=2, tile_rank=0, tile_world=world)
    fb = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ds.render_device(cam, p, fb.data_ptr(), stream)
    torch.cuda.synchronize()
    sync_ms = []
    for _ in range(3):
        ds.render_device(cam, p, fb.data_ptr(), stream)
        sync_ms.append(ds.last_kernel_ms())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        ds.render_device(cam, p, fb.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / k * 1e3
    print(f"1/{world} of C2: synchronous launches {min(sync_ms):.2f} ms (events), {k} back to back {ms:.2f} ms each (wall)", flush=True)
