"""Kernel time of every rank's share of an N-way tile partition on one GPU (are the shares equally dear?).
Usage (GPU box): python tools/experiments/rank_probe.py [C2|C4|C5|C3[:spp]] [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

CONFIGS = {"C2": ("random_spheres_iow", 1920, 1024), "C4": ("cornell_box", 1024, 4096), "C5": ("stress_spheres:500", 4096, 256),
           "C3": ("final_scene", 800, 10000)}
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene, width, spp = CONFIGS[name.split(":")[0]]
if ":" in name:
    spp = int(name.split(":")[1])
hs = HostScene(scene, 1)
cam = hs.next_camera()
ds = DeviceScene(hs.desc)
p = hs.params(width, spp, 50, seed=2)
ds.render(cam, p)
full = min(ds.render(cam, p)[1].kernel_ms for _ in range(2))
times = []
for r in range(n):
    p = hs.params(width, spp, 50, seed=2, tile_rank=r, tile_world=n)
    ds.render(cam, p)
    times.append(min(ds.render(cam, p)[1].kernel_ms for _ in range(2)))
print(f"{name}: whole frame {full:.2f} ms, ideal share {full / n:.2f} ms; ranks: " + " ".join(f"{t:.2f}" for t in times) +
      f"  max {max(times):.2f} mean {sum(times) / n:.2f}  efficiency of the slowest {full / n / max(times):.3f}", flush=True)
