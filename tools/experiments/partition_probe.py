"""What one rank of an N-GPU run costs: renders 1/N of a BASELINE config's tiles (tile_rank 0 of N) on one GPU and prints the
kernel time next to (full frame time)/N — the projected tile-parallel efficiency before the gather.
Usage: python tools/experiments/partition_probe.py [C2|C4|C5[:spp]] [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

CONFIGS = {"C2": ("random_spheres_iow", 1920, 1024), "C4": ("cornell_box", 1024, 4096), "C5": ("stress_spheres:500", 4096, 256),
           "C3": ("final_scene", 800, 10000)}
args = sys.argv[1:]
name = "C2"
if args and args[0].split(":")[0] in CONFIGS:
    name = args.pop(0)
scene, width, spp = CONFIGS[name.split(":")[0]]
if ":" in name:
    spp = int(name.split(":")[1])
hs = HostScene(scene, 1)
cam = hs.next_camera()
ds = DeviceScene(hs.desc)
full = None
for n in [1] + [int(a) for a in args]:
    p = hs.params(width, spp, 50, seed=2, tile_rank=0, tile_world=n)
    ds.render(cam, p)
    ms = min(ds.render(cam, p)[1].kernel_ms for _ in range(2 if full and full > 2000 else 3))
    if n == 1:
        full = ms
    print(f"{name} ({width} px, {spp} spp) 1/{n} of the tiles: kernel {ms:9.2f} ms   ideal {full / n:9.2f} ms   efficiency {full / n / ms:.3f}", flush=True)
