import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vecchio_amd import HostScene, DeviceScene, ffi
lib = ffi.load_device_lib()
lib.vk_debug_phase_stats.restype = C.c_int
lib.vk_debug_phase_stats.argtypes = [C.c_void_p, C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.POINTER(C.c_uint64 * 8)]
for name, w in (("final_scene", 400), ("random_spheres_demo", 640), ("perlin_demo", 640)):
    hs = HostScene(name, 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(w, 64, 50)
    ds.render(cam, p); img, st = ds.render(cam, p)
    out = (C.c_uint64 * 8)()
    rc = lib.vk_debug_phase_stats(ds._h, C.byref(cam), C.byref(p), C.byref(out))
    v = list(out); ns = p.width * p.height * p.samples_per_pixel
    print(f"{name}: {st.samples/st.kernel_ms/1e3:.1f} Msamples/s rc={rc}")
    if rc == 0:
        print(f"   per sample: box wave-steps {v[0]/ns:.3f} (fill {v[1]/max(1,v[0])/64:.3f}), prim phases {v[2]/ns:.3f} (fill {v[3]/max(1,v[2])/64:.3f}), shade phases {v[4]/ns:.3f} (fill {v[5]/max(1,v[4])/64:.3f}), rounds {v[6]/ns:.3f}; lane box {v[1]/ns:.1f} prim {v[3]/ns:.1f} shade {v[5]/ns:.1f}", flush=True)
    else:
        print("   ", lib.vk_last_error().decode())
