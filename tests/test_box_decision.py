"""AxisBB::hit (accel.rs:16-35) as the kernel evaluates it — reciprocal multiplies, one margin compare, the
reference's division sequence only inside the margin — must give the reference's boolean for EVERY box, ray and
tmax, including grazing rays, rays starting on a face, axis-parallel rays, zero/denormal/huge direction
components, infinite tmax and the always-hit leaf boxes.  Millions of cases through the kernel's own
box_step_core (tests/emu) against slab_exact alone."""
import ctypes as C

import numpy as np
import pytest


def decide(emu, boxes, rays, fused, perturb=0):
    """perturb != 0: the reciprocals 1/d moved by -1/0/+1 ulp per axis, as the device's v_rcp_f32 may deliver them"""
    lib = emu.load()
    lib.emu_box_decisions_rcp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_uint32]
    boxes = np.ascontiguousarray(boxes, np.float32)
    rays = np.ascontiguousarray(rays, np.float32)
    out = np.zeros(len(boxes), np.uint8)
    lib.emu_box_decisions_rcp(boxes.ctypes.data, rays.ctypes.data, len(boxes), out.ctypes.data, int(fused), int(perturb))
    return out


def make_boxes(rng, n, scale):
    c = rng.uniform(-scale, scale, (n, 3))
    h = rng.uniform(0.0, scale * 0.2, (n, 3)) * rng.choice([1.0, 1e-3, 0.0], (n, 3), p=[0.8, 0.15, 0.05])   # thin and flat boxes too
    b = np.empty((n, 6), np.float32)
    b[:, 0::2] = c - h
    b[:, 1::2] = c + h
    return b


@pytest.mark.parametrize("fused", [0, 1], ids=["sub_mul", "fused"])
def test_fast_path_equals_reference_divisions(emu, fused):
    """fused = 1: the fma(b, 1/d, -o/d) form of the sphere-only kernel variants (margin with an |o/d| term)"""
    rng = np.random.default_rng(7 + fused)
    n = 1_000_000
    total_fb = 0
    for scale, offset in ((1.0, 0.0), (500.0, 0.0), (1.0, 3.0e4)):
        # offset: the whole configuration far from the world origin — o/d is huge next to the t of the box, the worst case for
        # the fused form b*inv - o*inv (cancellation); its margin grows with |o/d| and the exact fallback takes over
        boxes = make_boxes(rng, n, scale) + np.float32(offset)
        o = rng.uniform(-scale, scale, (n, 3)).astype(np.float32) + np.float32(offset)
        # a third of the rays aim exactly at a point ON the box (corner, edge, face or interior point): grazing cases
        t = rng.uniform(0, 1, (n, 3)).astype(np.float32) * rng.choice([0.0, 1.0, 0.5], (n, 3)).astype(np.float32)
        target = boxes[:, 0::2] + t * (boxes[:, 1::2] - boxes[:, 0::2])
        d = (target - o).astype(np.float32) * rng.uniform(0.1, 3.0, (n, 1)).astype(np.float32)
        rnd = rng.normal(size=(n, 3)).astype(np.float32)
        pick = rng.uniform(size=n) < 0.35
        d[~pick] = rnd[~pick]
        # axis-parallel / degenerate components
        z = rng.uniform(size=(n, 3)) < 0.03
        d[z] = rng.choice(np.array([0.0, -0.0, 1e-38, 1e-31, 1e31, -1e-40], np.float32), int(z.sum()))
        # origins on a face
        onface = rng.uniform(size=n) < 0.1
        o[onface, 0] = boxes[onface, 0]
        tmax = rng.choice(np.array([np.inf, 1e30, 5.0, 0.5, 0.001, 0.0011], np.float32), n).astype(np.float32)
        near = rng.uniform(size=n) < 0.3            # tmax right at the box: |target - o| / |d| is where the ray reaches `target`
        with np.errstate(all="ignore"):
            tt = (np.linalg.norm(target - o, axis=1) / np.linalg.norm(d, axis=1)).astype(np.float32)
        tmax[near & np.isfinite(tt)] = tt[near & np.isfinite(tt)]
        rays = np.concatenate([o, d, tmax[:, None]], 1)
        dec = decide(emu, boxes, rays, fused)
        bad = ((dec & 1) != ((dec >> 1) & 1))
        assert not bad.any(), f"{int(bad.sum())} of {n} decisions differ, first: box {boxes[bad][0]} ray {rays[bad][0]}"
        # the device takes 1/d from v_rcp_f32 (1 ulp): the same cases with every reciprocal moved by -1 / 0 / +1 ulp
        for seed in (1, 2):
            decp = decide(emu, boxes, rays, fused, perturb=seed)
            badp = ((decp & 1) != ((decp >> 1) & 1))
            assert not badp.any(), f"rcp +-1 ulp: {int(badp.sum())} of {n} decisions differ, first: box {boxes[badp][0]} ray {rays[badp][0]}"
        total_fb += int(((dec >> 2) & 1).sum())
        assert ((dec & 1) == 1).mean() > 0.05 and ((dec & 1) == 0).mean() > 0.05      # both outcomes well covered
    assert 0 < total_fb < 0.2 * 3 * n          # the fallback is exercised, and is the exception


@pytest.mark.parametrize("fused", [0, 1], ids=["sub_mul", "fused"])
def test_always_hit_leaf_boxes(emu, fused):
    """objects that sit beside a BVH child are leaves with a +-3e38 box: hit for every ray the reference could form"""
    rng = np.random.default_rng(8)
    n = 200_000
    boxes = np.tile(np.array([-3e38, 3e38, -3e38, 3e38, -3e38, 3e38], np.float32), (n, 1))
    o = rng.uniform(-1e4, 1e4, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32) * rng.choice(np.array([1.0, 1e-6, 1e6], np.float32), (n, 1))
    tmax = rng.choice(np.array([np.inf, 10.0, 0.002], np.float32), n)
    dec = decide(emu, boxes, np.concatenate([o, d, tmax[:, None]], 1), fused)
    assert ((dec & 1) == ((dec >> 1) & 1)).all()
    assert (dec & 1).all()


def test_tmax_clamp_is_fminf_for_every_tmax_the_walk_can_hold(emu):
    """min_with_tmax (a signed-integer minimum of the bit patterns) against fminf, for tmax > 0 or +inf (never NaN: it starts at
    +inf and only ever takes accepted hit distances) and every kind of a: +-0, +-inf, denormals, negative, +NaN.  (-NaN is the one
    documented difference: it only arises when both bounds of an axis are NaN, i.e. on rays that take the exact test anyway.)"""
    lib = emu.load()
    lib.emu_min_with_tmax.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 1e-38, -1e-38, 3.4e38, -3.4e38, 0.001, 1.0, np.nan], np.float32)
    a = np.concatenate([special.repeat(64), rng.normal(size=200_000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 200_000).astype(np.float32),
                        rng.integers(0, 2**32, 200_000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    keep = ~(np.isnan(a) & (a.view(np.uint32) >> 31 == 1))          # drop -NaN bit patterns
    a = np.ascontiguousarray(a[keep])
    t = np.abs(rng.normal(size=len(a)).astype(np.float32)) * np.float32(10.0) ** rng.integers(-20, 20, len(a)).astype(np.float32)
    t[rng.uniform(size=len(a)) < 0.2] = np.inf
    t[t == 0] = np.float32(0.001)
    t = np.ascontiguousarray(t, np.float32)
    out = np.empty_like(a)
    lib.emu_min_with_tmax(a.ctypes.data, t.ctypes.data, len(a), out.ctypes.data)
    want = np.fmin(a, t)                                            # fminf: NaN operands are dropped
    assert (out.view(np.uint32) == want.view(np.uint32)).all()


def test_shared_reciprocal_division_is_exact_for_any_one_ulp_reciprocal(emu):
    """vk_trace.h div_by_a (the sphere test's quotients by |d|^2 on the device: v_rcp_f32, one Newton step, two fma corrections) is compiled
    for the device only — the ISA specifies v_rcp_f32 to 1 ulp, not to the bit — so the emulator divides.  The algorithm's claim is checked
    here instead: from ANY reciprocal within 1 ulp of 1/a the sequence ends on the correctly rounded n / a, over the operand ranges
    set_space admits.  (The GPU test test_gpu_parity.py compares the real instruction on 2^26 pairs.)"""
    import ctypes as C
    lib = emu.load()
    lib.emu_div_by_a_model.restype = C.c_uint64
    lib.emu_div_by_a_model.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_float * 4)]
    worst = (C.c_float * 4)()
    bad = lib.emu_div_by_a_model(4_000_000, 11, C.byref(worst))
    assert bad == 0, f"{bad} quotients differ, e.g. n={worst[0]!r} a={worst[1]!r}: got {worst[2]!r}, n/a = {worst[3]!r}"
