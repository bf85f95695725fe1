"""Longer runs of the gate lemma's two harnesses than the test suite affords (CPU; tests/test_gate_lemma.py holds the assertions):
    python tools/experiments/lemma_campaign.py [seeds] > profiles/rNN/lemma_campaign.log
A: the largest residual constant K = | |H - c| - R | * R / (u (rho + R)^2) over 5 M configurations per seed (plus hill climbing);
B: failures of the grown gates (must be 0) and of the bare ones, 4 M aimed rays per seed and setting."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_ffi  # noqa: E402

lib = emu_ffi.load()
lib.emu_lemma_residual.restype = C.c_double
lib.emu_lemma_residual.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
lib.emu_gate_soundness.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_float, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
worst = 0.0
for seed in range(100, 100 + seeds):
    k = lib.emu_lemma_residual(5_000_000, seed, 400_000)
    worst = max(worst, k)
    print(f"A seed {seed}: largest K {k:.3f}", flush=True)
print(f"A: largest K over {seeds} seeds x 5 M configurations: {worst:.3f} (used: 32)", flush=True)
tot = {}
for seed in range(200, 200 + seeds):
    for grow, pad, r0 in ((1, 0.25, 400.0), (1, 0.25, 4000.0), (0, 1.0 / 16.0, 400.0)):
        cnt = (C.c_uint64 * 2)(); v = (C.c_float * 19)()
        lib.emu_gate_soundness(4_000_000, seed, grow, pad, r0, cnt, v)
        key = ("grown" if grow else "bare", pad, r0)
        a = tot.setdefault(key, [0, 0]); a[0] += cnt[0]; a[1] += cnt[1]
    print(f"B seed {seed} done", flush=True)
for (kind, pad, r0), (cands, fails) in tot.items():
    print(f"B: {kind} gates, padding {pad:g}, ball {r0:g}: {fails} failures in {cands} accepted candidates", flush=True)
