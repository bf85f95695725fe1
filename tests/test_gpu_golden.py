"""-m gpu: the HIP path, through the C ABI, against the fixtures made from the reference's own sample
images (tests/golden/make_*.py) — the only reference-held artefacts for this path.
STATISTICAL by necessity: the reference is unseeded."""
import numpy as np
import pytest

import golden_checks as G
from vecchio_amd import DeviceScene

pytestmark = pytest.mark.gpu


def test_hip_cornell_matches_reference_png_statistical(device, host_scenes):
    """HEAD configuration exactly: cornell_box() (scene.rs:630-730), width 900 (main.rs:171), 1000 spp, depth 100
    (main.rs:28-29), PDF integrator, black background — 810 Msamples, rendered by the HIP path and compared with
    sample/therestofyourlife.png per 30x30-px block within the Monte-Carlo noise of the two images."""
    hs, cam = host_scenes("cornell_box")
    ds = DeviceScene(hs.desc)
    p = hs.params(900, 1000, 100)
    assert p.height == 900
    img, st = ds.render(cam, p)
    ds.close()
    assert np.isfinite(img).all()
    z, rel = G.cornell_blocks30_z(img)
    rms = float(np.sqrt((z ** 2).mean()))
    print(f"cornell 900x900x1000: {st.kernel_ms:.1f} ms; blocks {len(z)}; rms z {rms:.3f}; max |z| {np.abs(z).max():.2f}; "
          f"|z|>4: {(np.abs(z) > 4).mean():.4f}; median rel {np.median(rel):.4f}; max rel {rel.max():.4f}")
    assert rms < 1.5, f"block means disagree beyond Monte-Carlo noise: rms z = {rms:.2f}"
    assert (np.abs(z) > 4).mean() < 0.01
    assert np.abs(z).max() < 8
    assert np.median(rel) < 0.01
    # the 21-px border sees past the 555-box: background (0,0,0) exactly (main.rs:124)
    assert img[:5].max() == 0.0 and img[:, :5].max() == 0.0
    # Vec3::to_color of the image vs the PNG's codes, whole picture: mean code within a quarter of a code
    # (checks the output stage end to end against the reference's artefact)
    import json
    g6 = json.load(open(G.GOLDEN + "/cornell_blocks.json"))
    lin = ((G.to_color(img).astype(np.float64) + 0.5) / 256.0) ** 2
    m = lin.reshape(-1, 3).mean(0)
    assert (np.abs(m - np.array(g6["mean_linear_rgb"])) / np.array(g6["mean_linear_rgb"])).max() < 0.01


def test_hip_iow_regions_match_reference_png_statistical(device, host_scenes):
    """scatter integrator + sky + IOW camera + Metal/Lambertian/Dielectric of the HIP path against the
    scene-independent parts of sample/inoneweekend.png (1024x576): the pin for the C2 headline path."""
    hs, cam = host_scenes("random_spheres_iow")
    ds = DeviceScene(hs.desc)
    p = hs.params(1024, 256, 50)
    assert p.height == 576
    img, _ = ds.render(cam, p)
    ds.close()
    rep = G.iow_regions(img)
    print(rep)
    G.check_iow_regions(rep)


def test_hip_final_scene_matches_reference_nextweek_png_statistical(device, host_scenes):
    """final_scene() (scene.rs:732-874) at HEAD's settings — width 900 (main.rs:171), depth 100 (main.rs:29), PDF integrator —
    rendered by the HIP path at 2000 spp (1.6 G samples) against sample/thenextweek.png: fog haze, light silhouette,
    motion-blurred sphere, earth texels (the real earthmap), blue medium sphere, Perlin speckle statistics, fuzz-10 metal,
    the rotated cube of spheres.  The pin of everything C3 has and C4 has not (tests/golden/make_nextweek_regions.py)."""
    hs, cam = host_scenes("final_scene")
    ds = DeviceScene(hs.desc)
    p = hs.params(900, 2000, 100)
    assert p.height == 900
    img, st = ds.render(cam, p)
    ds.close()
    assert np.isfinite(img).all()
    rep = G.nextweek_regions(img)
    print(f"final_scene 900x900x2000: {st.kernel_ms:.0f} ms")
    for k, v in rep.items():
        print(k, {a: (np.round(b, 4).tolist() if not isinstance(b, int) else b) for a, b in v.items() if a not in ("mean", "ref")})
    G.check_nextweek_regions(rep, p.samples_per_pixel)
    # and the scatter integrator must NOT pass (see the CPU test of the same name)
    hs2, cam2 = host_scenes("final_scene_nextweek")
    ds2 = DeviceScene(hs2.desc)
    img2, _ = ds2.render(cam2, hs2.params(900, 256, 100))
    ds2.close()
    assert min(G.nextweek_regions(img2)["haze_upper_right"]["rel"]) > 0.2
