"""Pins the scatter/IOW path (C1/C2/C5's integrator, sky, camera; Metal, Lambertian::scatter, Dielectric)
to the reference's sample/inoneweekend.png through the parts of that picture that do not depend on its
unseeded random spheres (tests/golden/make_iow_regions.py).  STATISTICAL pin of the oracle; the HIP path is
checked against the same fixture in tests/test_gpu_golden.py."""
import golden_checks as G


def test_oracle_iow_regions_statistical(oracle, host_scenes):
    hs, cam = host_scenes("random_spheres_iow")
    p = hs.params(1024, 12, 50)
    assert p.height == 576
    img, _ = oracle.render(hs.desc, cam, p)
    rep = G.iow_regions(img)
    G.check_iow_regions(rep)
