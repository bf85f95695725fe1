"""Find samples in which a tree form differs from the tree as handed over on the emulator, and trace the first of them segment by segment.
usage: emu_diff_trace.py scene width spp [variant 0|1] [flags]"""
import os, sys, ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import emu_ffi
from vecchio_amd import HostScene, ffi

name, w, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
variant = sys.argv[4] if len(sys.argv) > 4 else "1"
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
os.environ["EMU_GLOBAL_VARIANT"] = variant
hs = HostScene(name, 1)
cam = hs.next_camera()
p = hs.params(w, spp, 50)
lib = emu_ffi.load()
lib.emu_sample.argtypes = [C.POINTER(ffi.SceneDesc), C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.c_uint32, C.c_uint32, C.POINTER(C.c_float * 3), C.POINTER(C.c_uint32)]
if len(sys.argv) > 6:
    pix, smp = int(sys.argv[6]), int(sys.argv[7])
    for fl in (2, flags):
        hs.desc.contents.flags = fl
        os.environ["EMU_TRACE"] = "1"
        out = (C.c_float * 3)(); dr = C.c_uint32()
        print(f"--- flags {fl}", file=sys.stderr, flush=True)
        lib.emu_sample(hs.desc, C.byref(cam), C.byref(p), pix, smp, C.byref(out), C.byref(dr))
        print(f"    rgb {list(out)} draws {dr.value}", file=sys.stderr, flush=True)
    sys.exit(0)
hs.desc.contents.flags = 2
_, ref, _, _ = emu_ffi.render_samples(hs.desc, cam, p)
hs.desc.contents.flags = flags
_, ps, _, _ = emu_ffi.render_samples(hs.desc, cam, p)
bad = np.argwhere((ps.view(np.uint32) != ref.view(np.uint32)).any(axis=1))[:, 0]
print(len(bad), "samples differ:", [(int(b) // spp, int(b) % spp) for b in bad[:20]])
