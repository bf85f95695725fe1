"""GPU-box helper (not a test): kernel-time throughput of the four BASELINE GPU configs at reduced spp + a sparse parity check
against the oracle, for one or more builds of the device library.
    python tools/experiments/perf_quick.py [--libs default,vecchio_amd/lib/exp/x.so] [--wl C2,C3,C4,C5] [--reps 2]
Each library is exercised in a child process (VK_DEVICE_LIB is read once per process)."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WL = {"C2": ("random_spheres_iow", 1920, 256), "C3": ("final_scene", 800, 1024), "C4": ("cornell_box", 1024, 1024), "C5": ("stress_spheres:500", 4096, 32)}


def child(wls, reps, check):
    import numpy as np
    import oracle_ffi as O
    from vecchio_amd import DeviceScene, HostScene
    out = {}
    for w in wls:
        name, width, spp = WL[w]
        hs = HostScene(name, 1)
        cam = hs.next_camera()
        ds = DeviceScene(hs.desc)
        if check:
            p = hs.params(width, 2, 50)
            img, _ = ds.render(cam, p)
            k = 997 if w != "C5" else 4099
            po = hs.params(width, 2, 50, tile_rank=5, tile_world=k)
            ref = np.full((p.height, p.width, 3), -1.0, np.float32)
            assert O.load().oracle_render(hs.desc, C.byref(cam), C.byref(po), ref.ctypes.data, 16, None) == 0
            m = ref[..., 0] >= 0
            err = float(np.abs(img[m] - ref[m]).max())
        else:
            err = None
        p = hs.params(width, spp, 50)
        best = 0.0
        for _ in range(reps):
            _, st = ds.render(cam, p)
            best = max(best, st.samples / st.kernel_ms / 1e3)
        out[w] = {"Msamples_per_s": round(best, 1), "err": err, "clamped": int(st.clamped_samples)}
        ds.close(); hs.close()
    print("RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--wl", default="C2,C3,C4,C5")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    wls = a.wl.split(",")
    if a.child:
        return child(wls, a.reps, not a.no_check)
    for lib in a.libs.split(","):
        env = dict(os.environ)
        env.pop("VK_DEVICE_LIB", None)
        if lib != "default":
            env["VK_DEVICE_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--wl", a.wl, "--reps", str(a.reps)] + (["--no-check"] if a.no_check else []),
                           env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(lib, "FAILED", r.stdout[-400:], r.stderr[-800:], flush=True)
            continue
        d = json.loads(line[0][7:])
        print(f"{os.path.basename(lib):28s} " + "  ".join(f"{w} {d[w]['Msamples_per_s']:8.1f} (err {d[w]['err'] if d[w]['err'] is None else format(d[w]['err'], '.1e')})" for w in wls), flush=True)


if __name__ == "__main__":
    main()
