"""GPU-box helper (not a test): throughput of the forms a sphere-only world can be walked in — the default (rebuilt tree with grown gates
where proven, else the tree as handed over), the empirical rebuilt tree (VK_SCENE_EMPIRICAL_TREES; VK_GATE_PROOF=0 forces it where the
proven one exists), the tree as handed over, VK_SCENE_FAST_ACCEL — each in a child process (the switches are read per scene / process).
    python tools/experiments/ab_forms.py [--wl C2,C5] [--spp 256,32] [--reps 2]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
WL = {"C2": ("random_spheres_iow", 1920), "C5": ("stress_spheres:500", 4096), "S100": ("stress_spheres:100", 2048)}


def child(wl, spp, flags, reps):
    from vecchio_amd import DeviceScene, HostScene, ffi
    name, width = WL[wl]
    hs = HostScene(name, 1)
    hs.desc.contents.flags = flags
    cam = hs.next_camera()
    ds = DeviceScene(hs.desc)
    p = hs.params(width, spp, 50)
    best, rq = 0.0, 0
    for _ in range(reps):
        _, st = ds.render(cam, p)
        best = max(best, st.samples / st.kernel_ms / 1e3)
        rq = ds.last_requeued_samples()
    print("RESULT " + json.dumps({"Msamples_per_s": round(best, 1), "requeued": rq, "samples": int(st.samples), "tree": int(ds.info().tree),
                                  "items": int(ds.info().n_items)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wl", default="C2,C5")
    ap.add_argument("--spp", default="256,32")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--child", nargs=3)
    a = ap.parse_args()
    if a.child:
        return child(a.child[0], int(a.child[1]), int(a.child[2]), a.reps)
    forms = [("default", 0, {}), ("empirical (r3 form)", 4, {"VK_GATE_PROOF": "0"}), ("handed over", 2, {}), ("fast accel", 1, {})]
    for wl, spp in zip(a.wl.split(","), a.spp.split(",")):
        for label, flags, env in forms:
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reps", str(a.reps), "--child", wl, spp, str(flags)], env=e,
                               capture_output=True, text=True, timeout=1200)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            print(f"{wl} x {spp} spp  {label:22s}", line[0][7:] if line else "FAILED " + r.stderr[-600:], flush=True)


if __name__ == "__main__":
    main()
