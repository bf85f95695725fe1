"""GPU-box helper (not a test): the final scene over several WORLD seeds (BVHNode::new's random axes put different objects into `len == 1`
nodes), every sample of the HIP path against the oracle.  Usage: python tools/experiments/dup_rule_seeds.py [first_seed n_seeds width spp]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O  # noqa: E402
from test_gpu_parity import compare_samples, device_samples  # noqa: E402
from vecchio_amd import DeviceScene, HostScene  # noqa: E402

first, n, width, spp = (int(x) for x in (sys.argv[1:5] + ["2", "10", "400", "16"][len(sys.argv) - 1:]))
total = 0
for name in ("final_scene", "bowser_demo", "random_spheres_demo"):
    for seed in range(first, first + n):
        hs = HostScene(name, seed)
        d = hs.desc.contents
        dups = {}
        for i in range(d.n_bvh):
            b = d.bvh[i]
            if b.left == b.right:
                k = (b.left >> 28) & 0xF
                dups[k] = dups.get(k, 0) + 1
        cam = hs.next_camera(); p = hs.params(width, spp, 50, seed=seed + 100)
        ds = DeviceScene(hs.desc)
        img_d, ps_d = device_samples(ds, cam, p)
        img_o, ps_o = O.render_samples(hs.desc, cam, p)
        compare_samples(ps_o, ps_d, img_o, img_d)
        total += ps_d.shape[0]
        print(f"{name} world seed {seed}: len-1 nodes by child kind {dups}; {ps_d.shape[0]} samples, every one on the oracle's path", flush=True)
        ds.close(); hs.close()
print(f"total {total} samples, all equal", flush=True)
