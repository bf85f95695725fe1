"""Closed-form unit tests of the oracle's intersectors and tie rules (the reference has no tests
of its own; these pin the restatement to the semantics read off hittable.rs / accel.rs)."""
import numpy as np
import pytest

from descs import Desc
from vecchio_amd import ffi


def test_sphere_front_hit(oracle):
    d = Desc()
    s = d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5))
    desc = d.finish(s)
    r = oracle.hit(desc, (0, 0, -5), (0, 0, 1))
    assert r is not None and r["t"] == pytest.approx(4.0) and r["front"]
    assert np.allclose(r["p"], (0, 0, -1)) and np.allclose(r["normal"], (0, 0, -1))
    # spherical(): phi = atan2(-1, 0) = -pi/2 -> u = 0.75, theta = asin(0) -> v = 0.5 (hittable.rs:54-61)
    assert r["u"] == pytest.approx(0.75) and r["v"] == pytest.approx(0.5)
    # direction is not normalised by the reference: t scales with 1/|d|
    r2 = oracle.hit(desc, (0, 0, -5), (0, 0, 2))
    assert r2["t"] == pytest.approx(2.0)


def test_sphere_inside_and_negative_radius(oracle):
    d = Desc()
    m = d.lambertian(0.5, 0.5, 0.5)
    desc = d.finish(d.sphere((0, 0, 0), 1.0, m))
    r = oracle.hit(desc, (0, 0, 0), (0, 0, 1))
    assert r["t"] == pytest.approx(1.0) and not r["front"] and np.allclose(r["normal"], (0, 0, -1))
    d2 = Desc()
    desc2 = d2.finish(d2.sphere((0, 0, 0), -1.0, d2.lambertian(0.5, 0.5, 0.5)))   # scene.rs:123-127 hollow glass
    r = oracle.hit(desc2, (0, 0, -5), (0, 0, 1))
    assert r["t"] == pytest.approx(4.0) and not r["front"] and np.allclose(r["normal"], (0, 0, -1))


def test_strict_vs_inclusive_bounds(oracle):
    d = Desc()
    desc = d.finish(d.xy_rect(-1, 1, -1, 1, 0.0, d.lambertian(0.5, 0.5, 0.5)))
    assert oracle.hit(desc, (0, 0, -4), (0, 0, 1), tmax=4.0) is not None     # Rect: t <= tmax is a hit (hittable.rs:232)
    assert oracle.hit(desc, (0, 0, -4), (0, 0, 1), tmin=4.0) is not None     # and t >= tmin
    assert oracle.hit(desc, (1.0, 1.0, -4), (0, 0, 1)) is not None           # extent bounds inclusive (hittable.rs:237)
    assert oracle.hit(desc, (1.0001, 0, -4), (0, 0, 1)) is None


def test_sphere_tmax_strict(oracle):
    d = Desc()
    desc = d.finish(d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5)))
    assert oracle.hit(desc, (0, 0, -5), (0, 0, 1), tmax=4.0) is None         # 4 < 4 false, 6 < 4 false (hittable.rs:75)
    assert oracle.hit(desc, (0, 0, -5), (0, 0, 1), tmax=4.0001)["t"] == pytest.approx(4.0)


def test_rect_uv_and_flipface(oracle):
    d = Desc()
    m = d.lambertian(0.5, 0.5, 0.5)
    rect = d.xz_rect(0, 4, 0, 2, 1.0, m)
    desc = d.finish(rect)
    r = oracle.hit(desc, (1, 5, 0.5), (0, -1, 0))
    assert r["t"] == pytest.approx(4.0) and r["u"] == pytest.approx(0.25) and r["v"] == pytest.approx(0.25)
    assert r["front"] and np.allclose(r["normal"], (0, 1, 0))
    d2 = Desc()
    rect2 = d2.xz_rect(0, 4, 0, 2, 1.0, d2.lambertian(0.5, 0.5, 0.5))
    desc2 = d2.finish(Desc.flip(rect2))
    r2 = oracle.hit(desc2, (1, 5, 0.5), (0, -1, 0))
    assert not r2["front"] and np.allclose(r2["normal"], (0, 1, 0))          # FlipFace negates front only (hittable.rs:299-308)


def test_tie_rules_bvh_right_list_first(oracle):
    # two coincident rects with different materials
    d = Desc()
    a = d.xy_rect(-1, 1, -1, 1, 0.0, d.lambertian(1, 0, 0))
    b = d.xy_rect(-1, 1, -1, 1, 0.0, d.lambertian(0, 1, 0))
    desc = d.finish(d.big_box(a, b))
    assert oracle.hit(desc, (0, 0, -3), (0, 0, 1))["material"] == 1          # BVH tie -> right (accel.rs:73-77)
    d = Desc()
    a = d.xy_rect(-1, 1, -1, 1, 0.0, d.lambertian(1, 0, 0))
    b = d.xy_rect(-1, 1, -1, 1, 0.0, d.lambertian(0, 1, 0))
    desc = d.finish(d.list_([a, b]))
    assert oracle.hit(desc, (0, 0, -3), (0, 0, 1))["material"] == 0          # list tie -> first (hittable.rs:386)


def test_boxy_and_translate_rotate(oracle):
    d = Desc()
    box = d.boxy((0, 0, 0), (2, 2, 2), d.lambertian(0.7, 0.7, 0.7))
    desc = d.finish(box)
    r = oracle.hit(desc, (1, 1, -5), (0, 0, 1))
    # the z = p0.z side is FlipFace(XYRect): outward normal +z faces away from the ray, so
    # set_face_normal gives front = false, normal = -z, and FlipFace turns front back to true
    assert r["t"] == pytest.approx(5.0) and np.allclose(r["normal"], (0, 0, -1)) and r["front"]
    d = Desc()
    s = d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5))
    desc = d.finish(d.translate(s, (1, 2, 3)))
    r = oracle.hit(desc, (1, 2, -5), (0, 0, 1))
    assert r["t"] == pytest.approx(7.0) and np.allclose(r["p"], (1, 2, 2)) and r["front"]
    d = Desc()
    rect = d.xy_rect(-1, 1, -1, 1, 2.0, d.lambertian(0.5, 0.5, 0.5))
    desc = d.finish(d.rotate(rect, 1, 90.0))                                  # RotateY(90): z=2 plane -> x=2 plane
    r = oracle.hit(desc, (5, 0, 0), (-1, 0, 0))
    assert r is not None and r["t"] == pytest.approx(3.0, abs=1e-4) and np.allclose(r["p"], (2, 0, 0), atol=1e-4)
    assert np.allclose(np.abs(r["normal"]), (1, 0, 0), atol=1e-5)


def test_moving_sphere(oracle):
    d = Desc()
    ms = d.moving_sphere((0, 0, 0), (10, 0, 0), 0.0, 1.0, 1.0, d.lambertian(0.5, 0.5, 0.5))
    desc = d.finish(ms)
    assert oracle.hit(desc, (5, 0, -5), (0, 0, 1), time=0.5)["t"] == pytest.approx(4.0)   # centre lerps with ray time
    assert oracle.hit(desc, (5, 0, -5), (0, 0, 1), time=0.0) is None


def test_constant_medium_transmittance(oracle):
    # P(scatter) inside a unit sphere of density 0.8 along a diameter = 1 - exp(-0.8 * 2)
    d = Desc()
    iso = d.mat(ffi.VK_MAT_ISOTROPIC, d.solid(1, 1, 1))
    b = d.sphere((0, 0, 0), 1.0, d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5))
    desc = d.finish(d.medium(b, 0.8, iso))
    n, hits, ts = 4000, 0, []
    for seed in range(n):
        r = oracle.hit(desc, (0, 0, -5), (0, 0, 1), seed=seed)
        if r is not None:
            hits += 1
            ts.append(r["t"])
            assert r["front"] and np.allclose(r["normal"], (1, 0, 0)) and r["material"] == iso
    want = 1 - np.exp(-1.6)
    assert abs(hits / n - want) < 3.5 * np.sqrt(want * (1 - want) / n)
    assert min(ts) >= 4.0 and max(ts) <= 6.0
