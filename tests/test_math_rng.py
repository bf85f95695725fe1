"""Shared host/device arithmetic (vecchio_amd/csrc/vk_math.h) checked against libm/numpy, and the
rand-0.7.3 draw mappings it restates.  CPU only."""
import numpy as np


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def check(got, want_f64, max_ulp=1, max_mismatch_frac=2e-6):
    want = want_f64.astype(np.float32)
    ok = np.isfinite(want)
    d = ulp_diff(got[ok], want[ok])
    assert d.max() <= max_ulp, f"max ulp {d.max()}"
    assert (d != 0).mean() <= max_mismatch_frac, f"mismatch fraction {(d != 0).mean()}"


def test_sin_cos(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-1e4, 1e4, 400000), rng.uniform(0, 2 * np.pi, 400000), np.linspace(-10, 10, 100001)]).astype(np.float32)
    check(oracle.math(0, x), np.sin(x.astype(np.float64)))
    check(oracle.math(1, x), np.cos(x.astype(np.float64)))
    assert np.isnan(oracle.math(0, np.array([np.inf, np.nan], np.float32))).all()


def test_log(oracle):
    rng = np.random.default_rng(2)
    u = (rng.integers(1, 1 << 24, 500000).astype(np.float32)) * np.float32(2.0 ** -24)   # the 24-bit draws of hittable.rs:473
    x = np.concatenate([u, rng.uniform(1e-30, 1e30, 200000).astype(np.float32)])
    check(oracle.math(2, x), np.log(x.astype(np.float64)))
    r = oracle.math(2, np.array([0.0, -1.0, np.inf], np.float32))
    assert r[0] == -np.inf and np.isnan(r[1]) and r[2] == np.inf


def test_asin_atan2(oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1, 1, 400000), [-1.0, 1.0, 0.0], 1 - np.logspace(-8, -1, 1000)]).astype(np.float32)
    check(oracle.math(3, x), np.arcsin(x.astype(np.float64)), max_ulp=1, max_mismatch_frac=1e-4)
    assert np.isnan(oracle.math(3, np.array([1.0000001, -1.5], np.float32))).all()       # hittable.rs:57 on |p.y| > 1
    y = rng.normal(size=400000).astype(np.float32)
    xx = rng.normal(size=400000).astype(np.float32)
    check(oracle.math(4, y, xx), np.arctan2(y.astype(np.float64), xx.astype(np.float64)))
    sp = oracle.math(4, np.array([0.0, 0.0, 1.0, -1.0], np.float32), np.array([1.0, -1.0, 0.0, 0.0], np.float32))
    assert np.allclose(sp, [0.0, np.pi, np.pi / 2, -np.pi / 2])


def test_pow5(oracle):
    x = np.random.default_rng(4).uniform(0, 1, 200000).astype(np.float32)
    check(oracle.math(5, x), x.astype(np.float64) ** 5)      # powf(5.0) of util.rs:28


def test_gen_f32_mapping(oracle):
    v = oracle.draws(7, 3, 5, 0, 200000)
    assert v.min() >= 0.0 and v.max() < 1.0
    assert np.all(v * 2 ** 24 == np.floor(v * 2 ** 24))          # 24-bit grid (rand 0.7.3 Standard)
    assert abs(v.mean() - 0.5) < 0.005 and abs(v.var() - 1 / 12) < 0.003
    assert np.array_equal(v, oracle.draws(7, 3, 5, 0, 200000))   # deterministic
    assert not np.array_equal(v[:100], oracle.draws(7, 3, 6, 0, 100))
    assert not np.array_equal(v[:100], oracle.draws(8, 3, 5, 0, 100))


def test_gen_range_mapping(oracle):
    v = oracle.draws(1, 0, 0, 1, 200000, lo=-1.0, hi=1.0)
    assert v.min() >= -1.0 and v.max() < 1.0 and abs(v.mean()) < 0.01
    v = oracle.draws(1, 0, 0, 1, 100000, lo=0.0, hi=float(np.float32(2 * np.pi)))
    assert v.min() >= 0.0 and v.max() < np.float32(2 * np.pi)
    v = oracle.draws(1, 0, 0, 1, 100000, lo=213.0, hi=343.0)     # Rect::random of the Cornell light (scene.rs:690-692)
    assert v.min() >= 213.0 and v.max() < 343.0 and abs(v.mean() - 278.0) < 1.0


def test_gen_index_mapping(oracle):
    for n in (1, 2, 3, 6, 7):
        v = oracle.draws(9, 1, 2, 2, 60000, n_index=n)
        assert v.min() >= 0 and v.max() <= n - 1
        counts = np.bincount(v.astype(np.int64), minlength=n) / len(v)
        assert np.abs(counts - 1.0 / n).max() < 0.01


def _chi2(counts):
    e = counts.sum() / counts.size
    return float(((counts - e) ** 2 / e).sum()), counts.size - 1


def test_sample_stream_statistics(oracle):
    """The per-sample generator (32-bit Weyl sequence + two-multiply finaliser keyed by the sample's
    64-bit key) is the build's own substitute for thread_rng(): uniform in 1, 2 and 3 dimensions
    within one stream, and across the streams of neighbouring (pixel, sample) pairs."""
    n = 1 << 21
    u = oracle.draws(11, 5, 9, 0, n).astype(np.float64)
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.004 and abs(np.corrcoef(u[:-2], u[2:])[0, 1]) < 0.004
    for counts in (np.histogram(u, bins=4096, range=(0, 1))[0],
                   np.histogram2d(u[:-1], u[1:], bins=96, range=((0, 1), (0, 1)))[0],
                   np.histogramdd(np.stack([u[0::3][:n // 3], u[1::3][:n // 3], u[2::3][:n // 3]], 1), bins=20, range=((0, 1),) * 3)[0]):
        c, dof = _chi2(counts)
        assert abs(c - dof) < 5.0 * np.sqrt(2.0 * dof), (c, dof)      # 5 sigma
    # first 8 draws of 40 000 neighbouring streams (the dimensions of camera jitter / lens / time)
    first = np.stack([oracle.draws(2, p, s, 0, 8) for p in range(625) for s in range(64)]).astype(np.float64)
    for a, b in ((0, 1), (0, 7), (2, 3)):
        c, dof = _chi2(np.histogram2d(first[:, a], first[:, b], bins=24, range=((0, 1), (0, 1)))[0])
        assert abs(c - dof) < 5.0 * np.sqrt(2.0 * dof), (a, b, c, dof)
    c, dof = _chi2(np.histogram2d(first[:-1, 0], first[1:, 0], bins=24, range=((0, 1), (0, 1)))[0])
    assert abs(c - dof) < 5.0 * np.sqrt(2.0 * dof), (c, dof)
    assert abs(np.corrcoef(first[:-1, 0], first[1:, 0])[0, 1]) < 0.02


def test_the_ranges_whose_redraw_test_never_fires():
    """vk_math.h gen_pm1 / gen_0_to restate gen_range(-1, 1) and gen_range(0, 2 pi) without rand 0.7.3's `if res < high` loop
    (UniformFloat::sample_single): every one of the 2^23 possible draws gives res < high, in f32 arithmetic as the code performs it
    (one multiply, one add, each rounded)."""
    k = np.arange(1 << 23, dtype=np.uint32)
    v01 = ((k | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)).astype(np.float32)
    pm1 = (v01 * np.float32(2.0)).astype(np.float32) + np.float32(-1.0)
    assert pm1.dtype == np.float32 and pm1.max() < np.float32(1.0) and pm1.min() == np.float32(-1.0)
    assert pm1.max() == np.float32(1.0 - 2.0 ** -22)
    two_pi = np.float32(2.0) * np.float32(3.14159265358979323846)
    a = (v01 * two_pi).astype(np.float32) + np.float32(0.0)
    assert a.max() < two_pi and a.min() == 0.0
