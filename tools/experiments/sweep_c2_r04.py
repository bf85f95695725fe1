"""GPU-box helper: scheduler thresholds of the sphere-only LDS kernels on C2 (VK_SHADE_DEFER x VK_PRIM_WEIGHT), one child per setting.
    python tools/experiments/sweep_c2_r04.py [spp]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spp = sys.argv[1] if len(sys.argv) > 1 else "512"
code = ("import sys; sys.path.insert(0, %r)\n"
        "from vecchio_amd import DeviceScene, HostScene\n"
        "hs = HostScene('random_spheres_iow', 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(1920, %s, 50)\n"
        "best = 0\n"
        "for _ in range(3):\n"
        "    _, st = ds.render(cam, p); best = max(best, st.samples / st.kernel_ms / 1e3)\n"
        "print('RATE', round(best, 1))\n") % (ROOT, spp)
for defer in ("4", "5", "6", "8", "12"):
    row = []
    for weight in ("1", "2", "3"):
        env = dict(os.environ, VK_SHADE_DEFER=defer, VK_PRIM_WEIGHT=weight)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        row.append(r.stdout.split("RATE")[-1].strip() if "RATE" in r.stdout else "FAIL")
    print(f"defer {defer}: weight 1/2/3 -> {' / '.join(row)}", flush=True)
