"""GPU box: random layer worlds (tests/test_retree.py LayerWorld) beyond the suite — the grid form through the C ABI against the oracle per
sample (equal draw counts, |dRGB| < 1e-4) and against VK_SCENE_REFERENCE_TREE bit for bit.  python tools/experiments/gpu_layer_fuzz.py [n first_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_ffi as O  # noqa: E402
from test_gpu_parity import compare_samples, device_samples  # noqa: E402
from test_retree import LayerWorld  # noqa: E402
from vecchio_amd import DeviceScene, ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
bad = 0; trees = {}
for k in range(n):
    seed = first + k
    desc, cam, p = LayerWorld(seed).build()
    img_o, ps_o = O.render_samples(desc, cam, p)
    out = []
    try:
        for flags in (0, ffi.VK_SCENE_REFERENCE_TREE):
            desc.contents.flags = flags
            ds = DeviceScene(desc)
            if flags == 0:
                t = int(ds.info().tree); trees[t] = trees.get(t, 0) + 1
            img_d, ps_d = device_samples(ds, cam, p); compare_samples(ps_o, ps_d, img_o, img_d); out.append(ps_d); ds.close()
        if not np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32)):
            bad += 1; print(f"layer {seed}: differs from the tree as handed over", flush=True)
    except AssertionError as e:
        bad += 1; print(f"layer {seed}: {str(e)[:200]}", flush=True)
    if (k + 1) % 50 == 0:
        print(f"{k + 1} worlds done, {bad} failures so far, tree forms {trees}", flush=True)
print("failures:", bad, "tree forms:", trees)
