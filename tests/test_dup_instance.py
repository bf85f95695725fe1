"""A `len == 1` node of BVHNode::new (accel.rs:108-111) holds the same object as both children and calls it twice, the second time
with tmax = the first call's hit.  The lineariser drops the second call where it cannot return anything: simple objects, and — since
round 4 — a Translate / Rotate chain over a child in which nothing draws (vk_linearize.cpp draw_free_instance; the final scene's sphere
cluster sits in such a node).  A chain over a ConstantMedium keeps both calls: the second one draws again (hittable.rs:473)."""
import numpy as np
import pytest

from descs import Desc, camera, params
from test_emu_parity import compare
from vecchio_amd import ffi


def _scene(with_medium, n=40, seed=7, bare=None):
    r = np.random.default_rng(seed)
    d = Desc()
    grey, glass = d.lambertian(0.6, 0.6, 0.7), d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5)
    iso = d.mat(ffi.VK_MAT_ISOTROPIC, d.solid(0.8, 0.8, 0.9))
    light = d.light(7, 7, 7)

    def tree(objs):
        if len(objs) == 1:
            return objs[0]
        k = len(objs) // 2
        (l, lb), (rr, rb) = tree(objs[:k]), tree(objs[k:])
        lo, hi = np.minimum(lb[0], rb[0]), np.maximum(lb[1], rb[1])
        return d.bvh_node(l, rr, tuple(lo.astype(np.float32)), tuple(hi.astype(np.float32))), (lo, hi)

    balls = []
    for _ in range(n):
        c, rad = r.uniform(-2, 2, 3), float(r.uniform(0.15, 0.5))
        balls.append((d.sphere(tuple(c), rad, grey if r.uniform() < 0.7 else glass), (c - rad, c + rad)))
    if with_medium:
        c = np.zeros(3)
        balls.append((d.medium(d.sphere((0.0, 0.0, 0.0), 1.5, glass), 0.8, iso), (c - 1.5, c + 1.5)))
    cluster, bb = tree(balls)
    if bare == "sphere":                # the chain's child is ONE object, not a BVH (DInstance::child_ref)
        cluster, bb = d.sphere((0.0, 0.0, 0.0), 1.2, glass), (np.full(3, -1.2), np.full(3, 1.2))
    elif bare == "medium":
        cluster, bb = d.medium(d.sphere((0.0, 0.0, 0.0), 1.5, glass), 0.8, iso), (np.full(3, -1.5), np.full(3, 1.5))
    inst = d.translate(d.rotate(cluster, 1, 15.0), (5.0, 5.0, 5.0))
    lo, hi = bb[0] - 1.0 + 5.0, bb[1] + 1.0 + 5.0
    if bare == "bvh":                   # the len-1 node's object is the BVHNode itself (a world list may hold one: scene.rs boxes1)
        inst, lo, hi = cluster, bb[0], bb[1]
    dup = d.bvh_node(inst, inst, tuple(lo.astype(np.float32)), tuple(hi.astype(np.float32)))          # the len == 1 node
    lref = d.xz_rect(2.0, 8.0, 2.0, 8.0, 9.5, light)
    world = d.bvh_node(dup, Desc.flip(lref), (-1.0, -1.0, -1.0), (11.0, 11.0, 11.0))
    desc = d.finish(world, [lref])
    return desc, camera((5.0, 5.5, -6.0), (5, 5, 5), vfov=45.0), params(40, 30, 8, max_depth=50, seed=11)


@pytest.mark.parametrize("with_medium", [False, True])
def test_a_len1_node_enters_a_draw_free_instance_once(with_medium, oracle, emu, built):
    desc, cam, p = _scene(with_medium)
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    assert info[2] == (2 if with_medium else 1), info      # instance records: one per call the device makes


def test_a_len1_node_over_a_bvh_walks_it_once(oracle, emu, built):
    sizes = {}
    for with_medium in (False, True):
        desc, cam, p = _scene(with_medium, bare="bvh")
        p = params(40, 30, 8, max_depth=50, seed=11)
        cam = camera((0.0, 0.5, -11.0), (0, 0, 0), vfov=45.0)
        img_o, ps_o = oracle.render_samples(desc, cam, p)
        img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
        compare(ps_o, ps_e, img_o, img_e)
        sizes[with_medium] = info[0]
    # 40 spheres: 50 items when the subtree is emitted once; with a ConstantMedium inside (41 objects) it is emitted twice
    assert sizes[False] <= 52 and sizes[True] >= 2 * sizes[False] - 4, sizes


@pytest.mark.parametrize("bare,records", [("sphere", 1), ("medium", 2)])
def test_a_len1_node_over_a_chain_with_one_object(bare, records, oracle, emu, built):
    desc, cam, p = _scene(False, bare=bare)
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    assert info[2] == records, info


@pytest.mark.gpu
@pytest.mark.parametrize("with_medium", [False, True])
def test_a_len1_node_on_the_gpu(with_medium, device, oracle):
    from test_gpu_parity import compare_samples, device_samples
    from vecchio_amd import DeviceScene
    desc, cam, p = _scene(with_medium)
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    ds = DeviceScene(desc)
    img_d, ps_d = device_samples(ds, cam, p)
    compare_samples(ps_o, ps_d, img_o, img_d)
    ds.close()


def test_the_final_scenes_cluster_is_entered_once(oracle, emu, host_scenes):
    """BASELINE's C3 world (scene seed 1): the 1 000-sphere cluster under Translate(RotateY) is alone in a slice of BVHNode::new, i.e. both
    children of a len-1 node.  One instance record, 173 instead of 213 steps per sample, every sample the oracle's (which makes both calls)."""
    hs, cam = host_scenes("final_scene")
    p = hs.params(48, 4, 50)
    img_o, ps_o = oracle.render_samples(hs.desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(hs.desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    assert info[2] == 1, info
    assert steps / ps_e.shape[0] < 195.0, steps / ps_e.shape[0]


@pytest.mark.parametrize("under_chain", [False, True])
def test_a_len1_node_over_coplanar_rects_is_entered_twice(under_chain, oracle, emu, built):
    """the "second call returns None" argument is Sphere::hit's strict `t < tmax` (hittable.rs:75).  Rect::hit accepts
    t == tmax (hittable.rs:232), so over a BVHNode of Rects the second call of a len-1 node does return hits — of whichever of several
    coplanar Rects its boxes still let through at tmax = t0 — and the lineariser must not drop it: a subtree that is not spheres only is
    emitted (or, under a Translate / Rotate chain, entered) twice, and every sample is the oracle's, which makes both calls."""
    r = np.random.default_rng(5)
    d = Desc()
    mats = [d.light(*c) for c in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1))]

    def tree(objs):
        if len(objs) == 1:
            return objs[0]
        k = len(objs) // 2
        (l, lb), (rr, rb) = tree(objs[:k]), tree(objs[k:])
        lo, hi = np.minimum(lb[0], rb[0]), np.maximum(lb[1], rb[1])
        return d.bvh_node(l, rr, tuple(lo.astype(np.float32)), tuple(hi.astype(np.float32))), (lo, hi)

    rects = []
    for i in range(12):                                     # overlapping quads in ONE plane (y = 1), six materials: every overlap is a tie
        x0, z0 = r.uniform(-2, 1, 2)
        w, h = r.uniform(0.8, 2.0, 2)
        ref = d.xz_rect(float(x0), float(x0 + w), float(z0), float(z0 + h), 1.0, mats[i % 6])
        rects.append((ref, (np.array([x0, 1.0 - 1e-4, z0]), np.array([x0 + w, 1.0 + 1e-4, z0 + h]))))
    sub, bb = tree(rects)
    lo, hi = bb
    if under_chain:
        sub = d.translate(sub, (0.5, 0.0, 0.25)); lo, hi = lo + [0.5, 0, 0.25], hi + [0.5, 0, 0.25]
    dup = d.bvh_node(sub, sub, tuple(lo.astype(np.float32)), tuple(hi.astype(np.float32)))
    floor = d.xz_rect(-5.0, 5.0, -5.0, 5.0, 3.0, d.lambertian(0.5, 0.5, 0.5))
    world = d.bvh_node(dup, floor, (-6.0, 0.0, -6.0), (6.0, 4.0, 6.0))
    desc = d.finish(world, [rects[0][0]])
    cam = camera((0.0, -4.0, 0.3), (0, 1, 0), vfov=50.0)
    p = params(48, 36, 6, max_depth=4, seed=3, integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    if under_chain:
        assert info[2] == 2, info                            # two instance records: the chain is entered twice
    else:
        assert info[1] == 2 * 12 + 1, info                   # primitive slots: the 12 Rects twice + the floor
