"""BASELINE config C1 (InOneWeekend random spheres 400x225, 100 spp, depth 50: the CPU plumbing case) through
the oracle, and multi-frame use of one uploaded scene (RotatingCamera, scene.rs:65-91)."""
import numpy as np
import pytest

from vecchio_amd import ffi


def test_c1_cpu_config(oracle, emu, host_scenes):
    hs, cam = host_scenes("random_spheres_iow")
    p = hs.params(400, 100, 50)
    assert (p.width, p.height) == (400, 225)
    img, cnt = oracle.render(hs.desc, cam, p)
    assert cnt.samples == 400 * 225 * 100 and cnt.n_dropped == 0
    assert np.isfinite(img).all() and img.min() >= 0.0
    # sky gradient at the top rows (blue > red), grey ground at the bottom, and the visit counts that
    # define the algorithmic bytes of this scene: ~103 box tests, ~14 sphere tests per sample
    top = img[-3:].mean(axis=(0, 1))
    assert top[2] > top[0] and top[2] > 0.8
    c = cnt.as_dict()
    assert 80 < c["n_aabb"] / c["samples"] < 130 and 8 < c["n_sphere"] / c["samples"] < 20
    # the kernel's formulation on a strip of the same frame (same seed): identical paths
    ps = hs.params(400, 2, 50, tile_rank=5, tile_world=400)
    a, _ = oracle.render(hs.desc, cam, ps)
    full2 = hs.params(400, 2, 50)
    b, _, _, _ = emu.render_samples(hs.desc, cam, full2)
    mask = a.sum(axis=2) > 0
    assert mask.sum() > 100 and np.abs(a[mask] - b[mask]).max() < 1e-5


@pytest.mark.gpu
def test_rotating_camera_frames_share_one_scene(device, oracle, built):
    from vecchio_amd import DeviceScene, HostScene
    hs = HostScene("random_spheres_demo", 1)     # cam_iter = RotatingCamera, 671 frames in the reference
    ds = DeviceScene(hs.desc)                     # uploaded once
    seen = []
    for frame in range(3):
        cam = hs.next_camera()
        assert cam is not None
        p = hs.params(96, 8, 50, seed=10 + frame)
        img, _ = ds.render(cam, p)
        ref, _ = oracle.render(hs.desc, cam, p)
        assert np.abs(img - ref).max() < 1e-4
        seen.append(img)
    assert not np.array_equal(seen[0], seen[1])
    ds.close()


@pytest.mark.gpu
def test_cli_harness_writes_reference_style_ppm(device, oracle, built, tmp_path):
    """vecchio_cli plays main(): scene -> BVH -> upload -> vk_render -> P3 PPM (main.rs:200-214)."""
    import os
    import subprocess
    from vecchio_amd import HostScene, build
    exe = build.build_cli()
    subprocess.check_call([exe, "cornell_box", "48", "8", "20", "1", "1"], cwd=tmp_path)
    toks = open(tmp_path / "output_0000.ppm").read().split()
    assert toks[0] == "P3" and toks[1:4] == ["48", "48", "255"]
    got = np.array(toks[4:], dtype=np.int64).reshape(48, 48, 3)
    hs = HostScene("cornell_box", 1)
    cam = hs.next_camera()
    p = hs.params(48, 8, 20, seed=2)               # the CLI renders with seed+1
    ref, _ = oracle.render(hs.desc, cam, p)
    want = np.zeros((48, 48, 3), np.uint8)
    ffi.load_host_lib().vkh_to_color(ref.ctypes.data, 48, 48, want.ctypes.data)
    assert np.abs(got - want.astype(np.int64)).max() <= 1     # 8-bit quantisation of values equal to 1e-6
    # a scene with an ImageTexture: the CLI finds the decoded copy of assets/earthmap.png through VECCHIO_ASSETS, and fails the way
    # the reference does (File::open(..).unwrap(), material.rs:270) when the directory does not hold it
    assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "assets")
    subprocess.check_call([exe, "final_scene", "40", "2", "8", "1", "1"], cwd=tmp_path, env=dict(os.environ, VECCHIO_ASSETS=assets))
    assert open(tmp_path / "output_0000.ppm").read().split()[1:4] == ["40", "40", "255"]
    r = subprocess.run([exe, "final_scene", "40", "2", "8", "1", "1"], cwd=tmp_path, env=dict(os.environ, VECCHIO_ASSETS=str(tmp_path)),
                       capture_output=True, text=True)
    assert r.returncode != 0 and "earthmap" in r.stderr


def test_to_color_restatements_agree_on_the_cpu(built):
    """Vec3::to_color (vec3.rs:44-61) three times over: the numpy restatement the GPU tests compare the device kernel with
    (tests/golden_checks.py), and the C++ host mirror's writer (vkh_to_color, what vecchio_cli prints) — on NaN, negatives,
    -0, values >= 1, +-inf, the 0.999 clamp boundary, code boundaries and random values, including the top-down row order."""
    import golden_checks as G
    from vecchio_amd import ffi
    rng = np.random.default_rng(5)
    w, h = 37, 11
    img = rng.uniform(0, 1.3, (h, w, 3)).astype(np.float32)
    b = np.float32(0.999) ** 2
    special = np.array([np.nan, -1.0, -0.0, 0.0, 1.0, 2.0, np.inf, -np.inf, b, np.nextafter(b, np.float32(0)), np.nextafter(b, np.float32(2)),
                        1e-45, 1e-38, (255.0 / 256) ** 2, (1.0 / 256) ** 2, 0.25, 1e30], np.float32)
    img.reshape(-1)[:len(special)] = special
    want = np.zeros((h, w, 3), np.uint8)
    ffi.load_host_lib().vkh_to_color(img.ctypes.data, w, h, want.ctypes.data)
    with np.errstate(invalid="ignore"):
        mine = G.to_color(img)[::-1]
    assert np.array_equal(mine, want)
    q = G.to_color(np.array([[[0.0, 1.0, 100.0]]], np.float32))
    assert q.tolist() == [[[0, 255, 255]]]
