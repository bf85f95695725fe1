"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TESTS ONLY — the product
package never imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

from vecchio_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "samples", "segments", "n_aabb", "n_sphere", "n_moving", "n_rect", "n_xform", "n_medium",
        "n_closest", "n_texel", "n_perlin", "n_draws", "n_dropped", "n_aabb_nonfinite", "n_sphere_nonfinite")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(ORACLE_SO)
    lib.oracle_render.restype = C.c_int
    lib.oracle_render.argtypes = [C.POINTER(ffi.SceneDesc), C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams),
                                  C.c_void_p, C.c_int, C.POINTER(Counters)]
    lib.oracle_sample.restype = C.c_int
    lib.oracle_sample.argtypes = [C.POINTER(ffi.SceneDesc), C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams),
                                  C.c_uint32, C.c_uint32, C.POINTER(C.c_float * 3), C.POINTER(C.c_uint32)]
    lib.oracle_hit.restype = C.c_int
    lib.oracle_hit.argtypes = [C.POINTER(ffi.SceneDesc), ffi.F3, ffi.F3, C.c_float, C.c_float, C.c_float, C.c_uint64,
                               C.POINTER(C.c_float * 11)]
    lib.oracle_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.oracle_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_size_t]
    lib.oracle_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def render(desc, cam, params, threads=None):
    lib = load()
    out = np.zeros((params.height, params.width, 3), dtype=np.float32)
    cnt = Counters()
    threads = threads or min(len(os.sched_getaffinity(0)), 32)
    st = lib.oracle_render(desc, C.byref(cam), C.byref(params), out.ctypes.data_as(C.c_void_p), threads, C.byref(cnt))
    if st != 0:
        raise RuntimeError(f"oracle status {st}: {lib.oracle_last_error().decode()}")
    return out, cnt


def render_samples(desc, cam, params, threads=None):
    """returns (image, per_sample[n,4]) — per_sample[:,3] is the draw count bit pattern"""
    lib = load()
    lib.oracle_render_samples.restype = C.c_int
    lib.oracle_render_samples.argtypes = [C.POINTER(ffi.SceneDesc), C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams),
                                          C.c_void_p, C.c_void_p, C.c_int]
    img = np.zeros((params.height, params.width, 3), dtype=np.float32)
    ps = np.zeros((params.width * params.height * params.samples_per_pixel, 4), dtype=np.float32)
    st = lib.oracle_render_samples(desc, C.byref(cam), C.byref(params), img.ctypes.data, ps.ctypes.data, threads or min(len(os.sched_getaffinity(0)), 32))
    if st != 0:
        raise RuntimeError(f"oracle status {st}: {lib.oracle_last_error().decode()}")
    return img, ps


def sample(desc, cam, params, pixel, s):
    lib = load()
    rgb = (C.c_float * 3)()
    draws = C.c_uint32()
    st = lib.oracle_sample(desc, C.byref(cam), C.byref(params), pixel, s, C.byref(rgb), C.byref(draws))
    if st != 0:
        raise RuntimeError(f"oracle status {st}: {lib.oracle_last_error().decode()}")
    return np.array(list(rgb), dtype=np.float32), draws.value


def hit(desc, origin, direction, time=0.0, tmin=0.001, tmax=float("inf"), seed=0):
    lib = load()
    rec = (C.c_float * 11)()
    h = lib.oracle_hit(desc, ffi.F3(*origin), ffi.F3(*direction), time, tmin, tmax, seed, C.byref(rec))
    if h < 0:
        raise RuntimeError(lib.oracle_last_error().decode())
    if not h:
        return None
    r = list(rec)
    return dict(p=r[0:3], normal=r[3:6], t=r[6], u=r[7], v=r[8], front=bool(r[9]), material=int(r[10]))


def math(op, a, b=None):
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
    out = np.empty_like(a)
    lib.oracle_math(op, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
    return out


def draws(seed, pixel, s, kind, n, lo=0.0, hi=1.0, n_index=1):
    lib = load()
    out = np.empty(n, dtype=np.float32)
    lib.oracle_draws(seed, pixel, s, kind, lo, hi, n_index, out.ctypes.data, n)
    return out
