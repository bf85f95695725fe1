"""Emulator A/B of the tree forms: box / sphere tests per sample and bit-equality with the tree as handed over.
usage: emu_forms.py [scene] [width] [spp]   (scene: random_spheres_iow | stress_spheres:N)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import emu_ffi
from vecchio_amd import HostScene, ffi

name = sys.argv[1] if len(sys.argv) > 1 else "random_spheres_iow"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 96
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
hs = HostScene(name, 1)
cam = hs.next_camera()
p = hs.params(w, spp, 50)
ref = None
for label, flags, env in (("handed over", 2, {}), ("default", 0, {}), ("fast accel", 1, {}), ("empirical", 4, {"VK_GATE_PROOF": "0"})):
    for variant in ("0", "1"):
        os.environ["EMU_GLOBAL_VARIANT"] = variant
        for k, v in env.items(): os.environ[k] = v
        hs.desc.contents.flags = flags
        emu_ffi.take_visit_counts(); emu_ffi.take_redo_stats()
        t0 = time.time()
        img, ps, steps, info = emu_ffi.render_samples(hs.desc, cam, p)
        nb, ns = emu_ffi.take_visit_counts(); redo, segs = emu_ffi.take_redo_stats()
        for k in env: del os.environ[k]
        n = p.width * p.height * spp
        if ref is None: ref = ps.copy()
        same = np.array_equal(ps.view(np.uint32), ref.view(np.uint32))
        print(f"{label:12s} {'global' if variant=='1' else 'lds   '}: items {info[0]:8d}  box {nb/n:8.2f} sphere {ns/n:6.2f} per sample, redo {redo} of {segs} segments, "
              f"{'== handed over' if same else 'DIFFERS in %d samples' % int((ps.view(np.uint32)!=ref.view(np.uint32)).any(axis=1).sum())}  ({time.time()-t0:.1f}s)", flush=True)
