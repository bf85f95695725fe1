"""The kernel's formulation (threaded stack-free BVH walk, deferred hit records, iterative
throughput integrator, reciprocal-multiply slab test with exact fallback) against the recursive
oracle, per SAMPLE, on the CPU: identical draw counts (= identical paths) and radiance equal to
re-association error.  Runs the same vk_trace.h the GPU compiles, via tests/emu."""
import numpy as np
import pytest

import special_scenes

BUILDER_SCENES = ["random_spheres_iow", "cornell_box", "final_scene", "random_spheres_demo", "perlin_demo", "balls_demo", "bowser_demo",
                  "random_spheres_iow+sah", "final_scene+sah", "cornell_box+sah"]     # +sah: SURVEY 8f-2 builder, same objects


def compare(ps_o, ps_e, img_o, img_e):
    d_o = ps_o[:, 3].view(np.uint32)
    d_e = ps_e[:, 3].view(np.uint32)
    assert np.array_equal(d_o, d_e), f"{int((d_o != d_e).sum())} samples took a different path (draw counts differ)"
    fo = np.isfinite(ps_o[:, :3]).all(1)
    fe = np.isfinite(ps_e[:, :3]).all(1)
    assert np.array_equal(fo, fe), "finite filter (main.rs:192-194) would drop different samples"
    a, b = ps_o[fo, :3], ps_e[fo, :3]
    rel = np.abs(a - b) / (np.abs(a) + 1e-3)
    assert rel.max() < 2e-5, f"per-sample radiance differs by {rel.max()}"
    # north_star tolerance on pixels (expected ~1e-6); relative for pixels brighter than 1 (a firefly of radiance 4 000 moves its pixel by
    # 1e-7 of that)
    assert (np.abs(img_o - img_e) / np.maximum(1.0, np.abs(img_o))).max() < 1e-4


@pytest.mark.parametrize("name", BUILDER_SCENES)
def test_builder_scene(name, oracle, emu, host_scenes):
    hs, cam = host_scenes(name)
    p = hs.params(40, 6, 50)
    img_o, ps_o = oracle.render_samples(hs.desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(hs.desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    assert steps > 0 and info[0] > 0


@pytest.mark.parametrize("name", sorted(special_scenes.ALL))
def test_special_scene(name, oracle, emu, built):
    d, desc, cam, p = special_scenes.ALL[name]()
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)


def test_other_seeds_and_depth_limit(oracle, emu, host_scenes):
    hs, cam = host_scenes("cornell_box")
    for seed, depth in ((11, 3), (12, 1), (13, 100)):
        p = hs.params(24, 8, depth, seed=seed)
        img_o, ps_o = oracle.render_samples(hs.desc, cam, p)
        img_e, ps_e, _, _ = emu.render_samples(hs.desc, cam, p)
        compare(ps_o, ps_e, img_o, img_e)


def test_stress_scene_subset(oracle, emu, built):
    """C5's generator at a reduced grid (deep BVH, 10K spheres)"""
    from vecchio_amd import HostScene
    hs = HostScene("stress_spheres:50", 1)
    cam = hs.next_camera()
    p = hs.params(32, 4, 50)
    img_o, ps_o = oracle.render_samples(hs.desc, cam, p)
    img_e, ps_e, _, info = emu.render_samples(hs.desc, cam, p)
    assert info[1] > 9000
    compare(ps_o, ps_e, img_o, img_e)


@pytest.mark.parametrize("poison", ["direction", "origin", "inf_origin"])
def test_non_finite_rays_in_a_sphere_only_scene(poison, oracle, emu, host_scenes):
    """A ray with a NaN direction or origin hits every box (f32::min/max drop NaN quotients, accel.rs:21-31) and no sphere
    (NaN discriminant, hittable.rs:66-70): the reference walks its whole tree and returns None.  In a spheres-only scene the
    kernel skips that walk (vk_trace.h begin_segment); the outcome — miss, NaN sky colour, sample dropped by the finite
    filter (main.rs:192-194) — and the draw count must be the oracle's."""
    import ctypes as C
    from vecchio_amd import ffi
    hs, cam0 = host_scenes("random_spheres_iow")
    cam = ffi.Camera.from_buffer_copy(cam0)
    if poison == "direction":
        cam.horizontal[1] = float("nan")
    elif poison == "origin":
        cam.origin[2] = float("nan")
    else:
        cam.origin[0] = float("inf")
    p = hs.params(24, 3, 50)
    img_o, ps_o = oracle.render_samples(hs.desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(hs.desc, cam, p)
    assert np.array_equal(ps_o[:, 3].view(np.uint32), ps_e[:, 3].view(np.uint32))
    assert np.array_equal(np.isfinite(ps_o[:, :3]).all(1), np.isfinite(ps_e[:, :3]).all(1))
    assert np.array_equal(img_o, img_e)
    if poison != "inf_origin":              # (an infinite origin gives direction (-inf, ..): unit(d).y = 0, a finite sky colour)
        assert not np.isfinite(ps_o[:, :3]).all(1).any() and img_o.max() == 0.0
    assert steps < 24 * 13 * 3 * 8          # no tree walk: the oracle's counters would say 511 box tests per sample
