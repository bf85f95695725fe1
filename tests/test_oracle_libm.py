"""An INDEPENDENT arithmetic check of the oracle.  vk_math.h's sin / cos / ln / asin / atan2 / x^5 (material.rs:51-58,
util.rs:52-63, hittable.rs:54-61,473, util.rs:25-29) are compiled into the oracle, the emulator AND the device, so an error in them would
be common-mode: the per-sample parity tests cannot see it.  oracle/_build/liboracle_libm.so is the same restatement with those seven
functions from glibc's f32 libm instead (oracle/Makefile `libm`).  The two oracles draw from the same streams, so a sample differs only
where a last-bit difference in a transcendental changes a decision or a radiance value:
  * draw counts differ (the sample took another PATH: another hit, rejection loop, scatter branch) in a tiny fraction of the samples;
  * the images agree far inside their Monte-Carlo error.
Numbers on BASELINE C1 (400 x 225 x 100 spp, depth 50) are printed; the bounds are ~10x what is measured."""
import ctypes as C
import os

import numpy as np
import pytest

from vecchio_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def render_samples_with(so, desc, cam, p):
    lib = C.CDLL(so)
    lib.oracle_render_samples.restype = C.c_int
    lib.oracle_render_samples.argtypes = [C.POINTER(ffi.SceneDesc), C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.c_void_p, C.c_void_p, C.c_int]
    img = np.zeros((p.height, p.width, 3), np.float32)
    ps = np.zeros((p.width * p.height * p.samples_per_pixel, 4), np.float32)
    assert lib.oracle_render_samples(desc, C.byref(cam), C.byref(p), img.ctypes.data, ps.ctypes.data, min(len(os.sched_getaffinity(0)), 32)) == 0
    return img, ps


@pytest.mark.parametrize("name,w,spp", [("random_spheres_iow", 400, 100), ("cornell_box", 200, 64), ("final_scene", 160, 32)])
def test_libm_oracle_agrees_with_the_shared_arithmetic(name, w, spp, built, host_scenes):
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    so_libm = os.path.join(ROOT, "oracle", "_build", "liboracle_libm.so")
    assert os.path.exists(so_libm), "oracle/Makefile builds it beside liboracle.so"
    hs, cam = host_scenes(name)
    p = hs.params(w, spp, 50, seed=2)
    img_a, ps_a = render_samples_with(so, hs.desc, cam, p)
    img_b, ps_b = render_samples_with(so_libm, hs.desc, cam, p)
    n = ps_a.shape[0]
    draws_differ = ps_a[:, 3].view(np.uint32) != ps_b[:, 3].view(np.uint32)
    rgb_differ = (ps_a[:, :3].view(np.uint32) != ps_b[:, :3].view(np.uint32)).any(axis=1)
    frac_path, frac_rgb = float(draws_differ.mean()), float(rgb_differ.mean())
    fin = np.isfinite(ps_a[:, :3]).all(axis=1) & np.isfinite(ps_b[:, :3]).all(axis=1)
    a, b = ps_a[fin, :3].astype(np.float64), ps_b[fin, :3].astype(np.float64)
    sigma = a.std(axis=0) / np.sqrt(len(a)) + 1e-12            # Monte-Carlo error of the image mean
    z = np.abs(a.mean(axis=0) - b.mean(axis=0)) / sigma
    # the two images, pixel by pixel: the same samples but for the few that moved
    dmax = float(np.abs(img_a - img_b).max())
    print(f"{name} {w} px x {spp} spp = {n} samples: another path (draw count differs) {frac_path:.2e}, another value {frac_rgb:.2e}; "
          f"image means differ by {z.max():.3f} sigma; largest pixel difference {dmax:.3g}")
    assert frac_path < 2e-3 and z.max() < 4.0
    # ... and every sample that did NOT change its path has (nearly) its value
    same_path = ~draws_differ & fin
    rel = np.abs(ps_a[same_path, :3] - ps_b[same_path, :3]) / (np.abs(ps_a[same_path, :3]) + 1e-3)
    assert float(rel.max()) < 1e-2 or float((rel > 1e-4).mean()) < 1e-3
