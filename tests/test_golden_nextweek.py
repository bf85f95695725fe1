"""Pins the oracle to sample/thenextweek.png (README.md:9-15): the reference's own render of final_scene()
(scene.rs:732-874) and the only reference-held artefact showing ConstantMedium + Isotropic, MovingSphere, Perlin /
NoiseTexture, ImageTexture (the real earthmap, tests/golden/assets/) and the fuzz-10 Metal.  STATISTICAL — the reference
is unseeded — through the regions tests/golden/make_nextweek_regions.py documents; the same check runs on the HIP path's
converged 900x900 render in tests/test_gpu_golden.py."""
import numpy as np

import golden_checks as G


def test_oracle_matches_reference_nextweek_regions_statistical(oracle, host_scenes):
    hs, cam = host_scenes("final_scene")
    p = hs.params(900, 12, 100)         # main.rs:171 width 900, MAX_DEPTH 100 (main.rs:29); 12 spp keeps the CPU suite short
    assert p.height == 900 and hs.integrator == 0          # HEAD's PDF integrator: see make_nextweek_regions.py
    img, cnt = oracle.render(hs.desc, cam, p)
    assert cnt.n_dropped < cnt.samples * 1e-3
    rep = G.nextweek_regions(img, quantise=False)
    print({k: np.round(v["rel"], 4).tolist() for k, v in rep.items() if "rel" in v})
    G.check_nextweek_regions(rep, p.samples_per_pixel)


def test_scatter_integrator_does_not_reproduce_nextweek_png(oracle, host_scenes):
    """The discriminating power of the check above: the same objects through a plain scatter integrator (emitted +
    attenuation * L, what the TheNextWeek TAG would have had) are tens of percent off the PNG in the fog — the PNG on master
    was rendered by the PDF integrator, and a wrong integrator / medium / phase function would not pass."""
    hs, cam = host_scenes("final_scene_nextweek")
    assert hs.integrator == 1
    p = hs.params(900, 2, 100)
    img, _ = oracle.render(hs.desc, cam, p)
    rep = G.nextweek_regions(img, quantise=False)
    assert min(rep["haze_upper_right"]["rel"]) > 0.2, rep["haze_upper_right"]
    assert min(rep["wall_mid"]["rel"]) > 0.2, rep["wall_mid"]


def test_earthmap_fixture_is_the_reference_asset():
    """tests/golden/assets/earthmap.ppm.gz decodes to the 1024x512 RGB8 buffer ImageTexture::new reads (material.rs:269-279)."""
    import gzip
    import hashlib
    import os
    raw = gzip.open(os.path.join(G.GOLDEN, "assets", "earthmap.ppm.gz")).read()
    assert raw.startswith(b"P6\n1024 512\n255\n")
    body = raw[len(b"P6\n1024 512\n255\n"):]
    assert len(body) == 1024 * 512 * 3
    assert hashlib.sha256(body).hexdigest().startswith("0651e147c9164cf9")     # printed by tests/golden/make_assets.py
