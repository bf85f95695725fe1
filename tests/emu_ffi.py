"""ctypes binding of tests/emu (host build of the kernel's per-lane logic).  TESTS ONLY."""
import ctypes as C
import os

import numpy as np

from vecchio_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "emu", "_build", "libemu.so")
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(EMU_SO):
        from vecchio_amd import build
        build.build_emu()
    lib = C.CDLL(EMU_SO)
    lib.emu_render.restype = C.c_int
    lib.emu_render.argtypes = [C.POINTER(ffi.SceneDesc), C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.c_void_p, C.c_void_p,
                               C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
    lib.emu_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def render_samples(desc, cam, p, threads=None):
    """returns (image, per_sample[n,4], steps, info) — per_sample[:,3] is the draw count bit pattern"""
    lib = load()
    img = np.zeros((p.height, p.width, 3), np.float32)
    ps = np.zeros((p.width * p.height * p.samples_per_pixel, 4), np.float32)
    steps = C.c_uint64()
    info = (C.c_uint32 * 4)()
    st = lib.emu_render(desc, C.byref(cam), C.byref(p), img.ctypes.data, ps.ctypes.data, threads or (os.cpu_count() or 1), C.byref(steps), info)
    if st != 0:
        raise RuntimeError(f"emu status {st}: {lib.emu_last_error().decode()}")
    return img, ps, steps.value, list(info)


def take_redo_stats():
    """exact re-treeing: (samples rendered again on the tree as handed over, segments walked) since the last call"""
    lib = load()
    out = (C.c_uint64 * 2)()
    lib.emu_take_redo_stats(out)
    return int(out[0]), int(out[1])


def take_visit_counts():
    """(box tests, sphere tests) of every walk since the last call — second walks and samples rendered again included; sphere tests are
    counted for scenes of spheres only (other scenes: primitive steps)"""
    lib = load()
    out = (C.c_uint64 * 2)()
    lib.emu_take_visit_counts(out)
    return int(out[0]), int(out[1])
