"""Quick GPU bring-up script (not a pytest): device vs oracle per-sample parity + a timing probe.
Usage: python tools/experiments/gpu_quick.py [scene ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from vecchio_amd import DeviceScene, HostScene, ffi  # noqa: E402
import oracle_ffi as O  # noqa: E402


def device_samples(ds, cam, p):
    lib = ffi.load_device_lib()
    lib.vk_debug_render_samples.restype = C.c_int
    lib.vk_debug_render_samples.argtypes = [C.c_void_p, C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.c_void_p, C.c_void_p]
    img = np.zeros((p.height, p.width, 3), np.float32)
    ps = np.zeros((p.width * p.height * p.samples_per_pixel, 4), np.float32)
    st = lib.vk_debug_render_samples(ds._h, C.byref(cam), C.byref(p), img.ctypes.data, ps.ctypes.data)
    if st != 0:
        raise RuntimeError(lib.vk_last_error().decode())
    return img, ps


def oracle_samples(hs, cam, p):
    ol = O.load()
    ol.oracle_render_samples.restype = C.c_int
    ol.oracle_render_samples.argtypes = [C.POINTER(ffi.SceneDesc), C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.c_void_p, C.c_void_p, C.c_int]
    img = np.zeros((p.height, p.width, 3), np.float32)
    ps = np.zeros((p.width * p.height * p.samples_per_pixel, 4), np.float32)
    st = ol.oracle_render_samples(hs.desc, C.byref(cam), C.byref(p), img.ctypes.data, ps.ctypes.data, os.cpu_count() or 1)
    assert st == 0, ol.oracle_last_error()
    return img, ps


def main():
    scenes = sys.argv[1:] or ["random_spheres_iow", "cornell_box", "final_scene", "random_spheres_demo", "perlin_demo", "balls_demo"]
    ok = True
    for name in scenes:
        hs = HostScene(name, 1)
        cam = hs.next_camera()
        p = hs.params(72, 8, 50)
        ds = DeviceScene(hs.desc)
        info = ds.info()
        t0 = time.time()
        img_d, ps_d = device_samples(ds, cam, p)
        t1 = time.time()
        img_o, ps_o = oracle_samples(hs, cam, p)
        d_o = ps_o[:, 3].view(np.uint32)
        d_d = ps_d[:, 3].view(np.uint32)
        mism = int((d_o != d_d).sum())
        fin = np.isfinite(ps_o[:, :3]).all(1) & np.isfinite(ps_d[:, :3]).all(1)
        nonfin = int((np.isfinite(ps_o[:, :3]).all(1) != np.isfinite(ps_d[:, :3]).all(1)).sum())
        rel = (np.abs(ps_o[fin, :3] - ps_d[fin, :3]) / (np.abs(ps_o[fin, :3]) + 1e-3)).max() if fin.any() else 0.0
        pix = np.abs(img_o - img_d).max()
        good = mism == 0 and nonfin == 0 and pix < 1e-4
        ok &= good
        print(f"{name}: items={info.n_items} lds={info.lds_bytes} feat={info.features:#x} samples={len(d_o)} draw_mismatch={mism} "
              f"nonfinite_mismatch={nonfin} max_rel_sample={rel:.3e} max_abs_pixel={pix:.3e} dev_time={t1 - t0:.3f}s {'OK' if good else 'FAIL'}", flush=True)
        if mism:
            bad = np.nonzero(d_o != d_d)[0][:5]
            for b in bad:
                print("   sample", b // p.samples_per_pixel, b % p.samples_per_pixel, "oracle", ps_o[b, :3], d_o[b], "device", ps_d[b, :3], d_d[b])
        ds.close()
    # timing probe on the headline scene
    hs = HostScene("random_spheres_iow", 1)
    cam = hs.next_camera()
    ds = DeviceScene(hs.desc)
    for (w, spp) in [(480, 16), (960, 64), (1920, 64)]:
        p = hs.params(w, spp, 50)
        img, st = ds.render(cam, p)
        print(f"timing {w}x{p.height} spp={spp}: kernel {st.kernel_ms:.2f} ms -> {st.samples / st.kernel_ms / 1e3:.1f} Msamples/s (lds={st.scene_in_lds})", flush=True)
    # phase scheduler statistics (instrumented kernel build)
    lib = ffi.load_debug_lib()
    lib.vk_debug_phase_stats.restype = C.c_int
    lib.vk_debug_phase_stats.argtypes = [C.c_void_p, C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.POINTER(C.c_uint64 * 24)]
    p = hs.params(1920, 64, 50)
    out = (C.c_uint64 * 24)()
    ds.close(); ds = DeviceScene(hs.desc, lib=lib)
    if lib.vk_debug_phase_stats(ds._h, C.byref(cam), C.byref(p), C.byref(out)) == 0:
        v = list(out)
        ns = p.width * p.height * p.samples_per_pixel
        print(f"phase stats per sample: box wave-steps {v[0]/ns:.3f} (lane fill {v[1]/max(1,v[0])/64:.3f}), prim phases {v[2]/ns:.3f} (fill {v[3]/max(1,v[2])/64:.3f}), "
              f"shade phases {v[4]/ns:.3f} (fill {v[5]/max(1,v[4])/64:.3f}), rounds {v[6]/ns:.3f}; lane box steps/sample {v[1]/ns:.1f} prim {v[3]/ns:.1f} shade+need {v[5]/ns:.1f}", flush=True)
    else:
        print("phase stats failed:", lib.vk_last_error().decode())
    print("ALL OK" if ok else "SOME FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
