"""CPU (emulator) campaign beyond the test suite: random layer worlds (the grid form) and sphere crowds (the tree forms), every sample against
the tree as handed over, bit for bit; the grid walk against all spheres.  python tools/experiments/emu_fuzz_more.py [n_layer n_crowd first_seed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import emu_ffi as E  # noqa: E402
from test_retree import LayerWorld, SphereCrowd  # noqa: E402
from vecchio_amd import ffi  # noqa: E402

n_layer = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_crowd = int(sys.argv[2]) if len(sys.argv) > 2 else 200
first = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
lib = E.load()
lib.emu_grid_claims.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
bad = 0
for kind, n, gen in (("layer", n_layer, LayerWorld), ("crowd", n_crowd, SphereCrowd)):
    for k in range(n):
        seed = first + k
        desc, cam, p = (gen(seed).build() if kind == "layer" else gen(seed, nasty=k % 3 == 0).build())
        desc.contents.flags = ffi.VK_SCENE_REFERENCE_TREE
        ref = E.render_samples(desc, cam, p)[1].view(np.uint32)
        desc.contents.flags = 0
        if kind == "layer":
            cnt = (C.c_uint64 * 3)(); v = (C.c_float * 8)()
            for key in ("EMU_GRID", "EMU_GLOBAL_VARIANT"):
                os.environ.pop(key, None)
            rc = lib.emu_grid_claims(desc, 5000, seed, cnt, v)
            if rc != 0 or cnt[2] != 0:
                bad += 1; print(f"{kind} {seed}: grid claims rc {rc} {list(cnt)} {list(v)}", flush=True)
        for env in ({}, {"EMU_GRID": "0"}, {"EMU_GRID": "0", "EMU_GLOBAL_VARIANT": "1"}):
            for key in ("EMU_GRID", "EMU_GLOBAL_VARIANT"):
                os.environ.pop(key, None)
            os.environ.update(env)
            got = E.render_samples(desc, cam, p)[1].view(np.uint32)
            nd = int((got != ref).any(axis=1).sum())
            if nd:
                bad += 1; print(f"{kind} {seed} {env}: {nd} samples differ", flush=True)
        if (k + 1) % 50 == 0:
            print(f"{kind}: {k + 1} done, {bad} failures so far", flush=True)
print("failures:", bad)
