import os, subprocess, sys, json
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
code = ("import sys; sys.path.insert(0, %r)\n"
        "from vecchio_amd import DeviceScene, HostScene, ffi\n"
        "hs = HostScene('random_spheres_iow', 1); hs.desc.contents.flags = int(sys.argv[1]); cam = hs.next_camera(); ds = DeviceScene(hs.desc); p = hs.params(1920, 1024, 50)\n"
        "best = 0\n"
        "for _ in range(3):\n"
        "    _, st = ds.render(cam, p); best = max(best, st.samples / st.kernel_ms / 1e3)\n"
        "print('RATE', round(best, 1), 'tree', ds.info().tree, 'requeued', ds.last_requeued_samples())\n") % ROOT
for rep in range(2):
    for label, flags, env in (("proven pad 1/4", 0, {}), ("empirical pad 1/16", 4, {"VK_GATE_PROOF": "0"}), ("empirical pad 1/4", 4, {"VK_GATE_PROOF": "0", "VK_T_PAD": "0.25"}),
                              ("proven pad 3/16", 0, {"VK_T_PAD": "0.1875"})):
        r = subprocess.run([sys.executable, "-c", code, str(flags)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        print(label, r.stdout.strip().split("\n")[-1] if r.stdout else r.stderr[-300:], flush=True)
