"""What one rank of an N-GPU run costs: renders 1/N of C2's tiles (tile_rank 0 of N) on one GPU and prints the
kernel time next to (full frame time)/N.  Usage: python tests/partition_probe.py [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

hs = HostScene("random_spheres_iow", 1)
cam = hs.next_camera()
ds = DeviceScene(hs.desc)
full = None
for n in [1] + [int(a) for a in sys.argv[1:]]:
    p = hs.params(1920, 1024, 50, seed=2, tile_rank=0, tile_world=n)
    ds.render(cam, p)
    ms = min(ds.render(cam, p)[1].kernel_ms for _ in range(3))
    if n == 1:
        full = ms
    print(f"1/{n} of the tiles: kernel {ms:8.2f} ms   ideal {full / n:8.2f} ms   efficiency {full / n / ms:.3f}", flush=True)
