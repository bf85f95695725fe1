"""GPU-box helper for profiling: build a scene, render it N times through the C ABI, print the kernel time.  No oracle, no bench logic.
    python tools/experiments/render_once.py <scene> <width> <spp> [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vecchio_amd import DeviceScene, HostScene

name, width, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
hs = HostScene(name, 1)
cam = hs.next_camera()
ds = DeviceScene(hs.desc)
p = hs.params(width, spp, 50)
for _ in range(reps):
    img, st = ds.render(cam, p)
    print(f"{name} {width}x{p.height}x{spp}: kernel {st.kernel_ms:.2f} ms, {st.samples / st.kernel_ms / 1e3:.1f} Msamples/s", flush=True)
ds.close()
