"""Phase-scheduler statistics of the instrumented kernel builds (GPU box): python tools/experiments/phase_stats.py [scene:width:spp ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from vecchio_amd import HostScene, DeviceScene, ffi
lib = ffi.load_debug_lib()
lib.vk_debug_phase_stats.restype = C.c_int
lib.vk_debug_phase_stats.argtypes = [C.c_void_p, C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.POINTER(C.c_uint64 * 24)]
jobs = [a.split(":") for a in sys.argv[1:]] or [("final_scene", 800, 256), ("random_spheres_iow", 1920, 128)]
for job in jobs:
    name, w, spp = ":".join(job[:-2]), int(job[-2]), int(job[-1])
    hs = HostScene(name, 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc, lib=lib); p = hs.params(w, spp, 50)
    ds.render(cam, p); img, st = ds.render(cam, p)
    out = (C.c_uint64 * 24)()
    rc = lib.vk_debug_phase_stats(ds._h, C.byref(cam), C.byref(p), C.byref(out))
    v = list(out); ns = p.width * p.height * p.samples_per_pixel
    print(f"{name}: {st.samples/st.kernel_ms/1e3:.1f} Msamples/s rc={rc} {lib.vk_last_error().decode() if rc else ''}")
    if rc == 0:
        tot = max(1, v[12])
        print(f"   per sample: box wave-steps {v[0]/ns:.3f} (fill {v[1]/max(1,v[0])/64:.3f}; lane box steps {v[1]/ns:.1f}), prim phases {v[2]/ns:.3f} (heavy {v[7]/ns:.3f}, fill {v[3]/max(1,v[2])/64:.3f}), shade phases {v[4]/ns:.3f} (fill {v[5]/max(1,v[4])/64:.3f}) rounds {v[6]/ns:.3f}")
        print(f"   wave clocks: box {v[8]/tot:.3f} light {v[9]/tot:.3f} heavy {v[10]/tot:.3f} shade {v[11]/tot:.3f} | clocks per: box step {v[8]/max(1,v[0]):.0f}, light prim {v[9]/max(1,v[2]-v[7]):.0f}, heavy prim {v[10]/max(1,v[7]):.0f}, shade phase {v[11]/max(1,v[4]):.0f}", flush=True)
        if ds.info().features == 0:
            print(f"   sphere-only builds: per exit test of the box loop: live {v[7]/max(1,v[2]):.1f} prim-pending {v[9]/max(1,v[2]):.1f} waiting-for-shade {v[10]/max(1,v[2]):.1f} (exit tests {v[2]/ns:.2f}/sample)")
        print(f"   shade phase split (clocks per phase): record+material {v[13]/max(1,v[4]):.0f}, refill {v[14]/max(1,v[4]):.0f}, install+cold store {v[15]/max(1,v[4]):.0f}, cooperative turbulence {v[16]/max(1,v[4]):.0f}, cold load {v[17]/max(1,v[4]):.0f}, rest {(v[11]-v[13]-v[14]-v[15]-v[16]-v[17])/max(1,v[4]):.0f}", flush=True)
    ds.close()
