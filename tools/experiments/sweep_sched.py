"""GPU-box helper (not a test): kernel-time throughput of one scene over the scheduler thresholds VK_SHADE_DEFER x VK_PRIM_WEIGHT (read per
render call by the library).    python tools/experiments/sweep_sched.py final_scene:800:512 4,6,8,12 1,2,3"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
job = sys.argv[1].split(":"); name, w, spp = ":".join(job[:-2]), int(job[-2]), int(job[-1])
defers = [int(x) for x in sys.argv[2].split(",")]; weights = [int(x) for x in sys.argv[3].split(",")]
import subprocess, json
if len(sys.argv) > 4 and sys.argv[4] == "--child":
    from vecchio_amd import HostScene, DeviceScene
    hs = HostScene(name, 1); cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(w, spp, 50)
    ds.render(cam, p); best = 0.0
    for _ in range(2):
        _, st = ds.render(cam, p); best = max(best, st.samples / st.kernel_ms / 1e3)
    print("RESULT", round(best, 1), flush=True)
    sys.exit(0)
for dfr in [0] + defers:
    row = []
    for pw in ([0] if dfr == 0 else weights):
        env = dict(os.environ)
        if dfr: env["VK_SHADE_DEFER"] = str(dfr); env["VK_PRIM_WEIGHT"] = str(pw)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:4] + ["--child"], env=env, capture_output=True, text=True)
        v = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("RESULT")]
        row.append(v[0] if v else "fail")
    print(f"{name} shade_defer={dfr or 'default'}: " + "  ".join(f"pw{pw or '-'}={x}" for pw, x in zip(([0] if dfr == 0 else weights), row)), flush=True)
