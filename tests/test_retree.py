"""Re-treeing (vk_linearize.cpp; opt-in, vk_scene_desc.flags & VK_SCENE_FAST_ACCEL): the lineariser rebuilds draw-free subtrees (>= 16 Sphere / Rect / Boxy / list objects, no
ConstantMedium, no Translate/Rotate) with a SAH builder, because BVHNode::hit's result (accel.rs:58-83) does not depend on
the tree over such objects — except for EXACT ties in t, where the reference's choice (Rect::hit's inclusive bound
hittable.rs:232, Sphere::hit's strict one :75, the list scan's strict one :386, BVHNode::hit's `l.t < r.t` accel.rs:73)
is reproduced from the objects' positions in the reference's visiting order.  These scenes are crowds of such objects with
deliberately coincident geometry (every hit of a duplicated object is an exact tie between two materials), in random tree
shapes, and the kernel's formulation must agree with the recursive oracle on every sample — with and without re-treeing.

Exact re-treeing (the default for worlds of spheres only; vk_trace.h segment_unsafe) is tested further down:
SphereCrowd worlds, the stress scenes, the headline scene — every sample must be the handed-over tree's, bit for bit, in both of the
device's forms (whole samples rendered again for scenes staged in LDS, segments walked again in place for scenes in global memory)."""
import os

import numpy as np
import pytest

from descs import Desc, camera, params
from test_emu_parity import compare
from test_fuzz_scenes import Gen, _union
from vecchio_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Crowd(Gen):
    def crowd(self, n):
        objs = []
        while len(objs) < n:
            k = self.r.uniform()
            if k < 0.45:
                ref, bb = self.sphere()
            elif k < 0.7:
                ref, bb = self.rect()
            elif k < 0.85:
                ref, bb = self.boxy()
            else:                                           # a generic list of spheres / rects (no moving spheres: those are not re-treed)
                items, bb = [], None
                for _ in range(int(self.r.integers(1, 4))):
                    r_, b = (self.sphere if self.r.uniform() < 0.5 else self.rect)()
                    items.append(Desc.flip(r_) if self.r.uniform() < 0.3 else r_); bb = b if bb is None else _union(bb, b)
                ref = self.d.list_(items)
            if self.r.uniform() < 0.2:
                ref = Desc.flip(ref)
            objs.append((ref, bb))
            # coincident copies with another material: same geometry, so every hit of it is an exact tie
            if self.r.uniform() < 0.35:
                kind, idx = ref >> 28, ref & ffi.VK_REF_INDEX_MASK
                if kind == ffi.VK_KIND_SPHERE:
                    s = self.d.spheres[idx]
                    objs.append((self.d.sphere(tuple(s.center), s.radius, self.mat()), bb))
                elif kind == ffi.VK_KIND_RECT:
                    q = self.d.rects[idx]
                    dup = self.d.rect(q.c0, q.c1, q.d0, q.d1, q.k, (q.axis0, q.axis1, q.axis2), self.mat())
                    objs.append((dup, bb))
                    if self.r.uniform() < 0.5:              # ... and a Boxy whose face lies in the same plane
                        lo, hi = np.array(bb[0]) + 1e-3, np.array(bb[1]) - 1e-3
                        lo[q.axis2] = q.k; hi[q.axis2] = q.k + 1.0
                        objs.append((self.d.boxy(tuple(lo), tuple(hi), self.mat()), (lo - 1e-3, hi + 1e-3)))
                if self.r.uniform() < 0.3 and len(objs) >= 2:
                    objs.append(objs[-2])                   # the very same object twice in the tree (a shared Arc)
        return objs

    def build(self):
        n = int(self.r.integers(18, 56))
        objs = self.crowd(n)
        order = self.r.permutation(len(objs))
        ref, bb = self.tree([objs[i] for i in order])
        top = [(ref, bb)]
        if self.r.uniform() < 0.5:                          # the crowd under a transform (an instance's own item range)
            off = self.r.uniform(-1, 1, 3)
            top = [(self.d.translate(ref, tuple(off)), (bb[0] + off, bb[1] + off))]
        if self.r.uniform() < 0.6:                          # something that draws during traversal next to it: only the crowd is re-treed
            top.append(self.medium())
        if self.r.uniform() < 0.5:
            top.append(self.moving())
        lref, lbb = self.rect(m=self.emit)
        top.append((Desc.flip(lref), lbb)); self.lights.append(lref)
        world, _ = self.tree([top[i] for i in self.r.permutation(len(top))])
        if (world >> 28) != ffi.VK_KIND_BVH:
            world = self.d.big_box(world, world)
        desc = self.d.finish(world, self.lights if self.use_pdf else [])
        desc.contents.flags = ffi.VK_SCENE_FAST_ACCEL
        ang = self.r.uniform(0, 6.28)
        cam = camera((5 + 13 * np.cos(ang), self.r.uniform(3, 8), 5 + 13 * np.sin(ang)), (5, 5, 5), vfov=50.0)
        kw = {} if self.use_pdf else dict(integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)
        return desc, cam, params(24, 20, 4, max_depth=int(self.r.choice([3, 12, 50])), seed=int(self.r.integers(1, 1000)), **kw)


@pytest.mark.parametrize("seed", range(24))
def test_retreed_crowd_matches_oracle_per_sample(seed, oracle, emu, built):
    desc, cam, p = Crowd(5000 + seed).build()
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    # the same scene on the reference's own tree: same samples, more steps
    desc.contents.flags = 0
    img_r, ps_r, steps_r, info_r = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_r, img_o, img_r)
    assert info_r[0] != info[0] or steps_r == steps     # a different item array unless nothing qualified


def test_retree_cuts_the_box_tests_of_the_headline_scene(emu, built):
    from vecchio_amd import HostScene
    hs = HostScene("random_spheres_iow", 1)
    cam = hs.next_camera()
    p = hs.params(64, 2, 50)
    hs.desc.contents.flags = ffi.VK_SCENE_REFERENCE_TREE
    _, ps_r, steps_r, info_r = emu.render_samples(hs.desc, cam, p)          # the tree handed over
    hs.desc.contents.flags = ffi.VK_SCENE_FAST_ACCEL
    _, ps, steps, info = emu.render_samples(hs.desc, cam, p)
    assert np.array_equal(ps.view(np.uint32), ps_r.view(np.uint32))      # bit-identical samples
    assert info_r[0] == 511 and steps < 0.7 * steps_r
    hs.desc.contents.flags = 0                                           # the default: exact re-treeing (a scene of spheres only)
    _, ps_x, steps_x, _ = emu.render_samples(hs.desc, cam, p)
    assert np.array_equal(ps_x.view(np.uint32), ps_r.view(np.uint32))
    assert steps_x < 0.8 * steps_r


@pytest.mark.parametrize("form,variant", [("empirical", "lds"), ("empirical", "global"), ("near", "global"), ("near", "global-no-primary-ref"),
                                          ("grid", "lds")])
@pytest.mark.parametrize("grid_half", [40, 130])
def test_exact_retree_gives_the_handed_over_trees_samples_on_far_small_spheres(grid_half, form, variant, oracle, emu, built, monkeypatch):
    """The default for a scene of spheres only (vk_trace.h segment_unsafe): a tree rebuilt over the reference's leaf units; where the
    winner of a segment could depend on the visiting order, the tree as handed over decides — for the whole sample (scenes the device
    stages in LDS: a second launch) or for that segment (scenes it traverses from global memory: both trees in one array).  Thousands
    of pixel-sized spheres on a ground sphere of radius 1e5 are where computed hits precede their box entries: every sample must still
    be the handed-over tree's, bit for bit, and the oracle's within the usual tolerance."""
    import emu_ffi
    from vecchio_amd import HostScene
    monkeypatch.setenv("EMU_GLOBAL_VARIANT", "1" if variant.startswith("global") else "0")
    hs = HostScene(f"stress_spheres:{grid_half}", 1)
    cam = hs.next_camera()
    p = hs.params(72, 2, 50, seed=5)
    img_o, ps_o = oracle.render_samples(hs.desc, cam, p)                    # the recursive restatement on the tree handed over
    # BVHNode::new's units are long on this scene: grown UNIT gates would be too dear.  The default is the NEAR form (round 5: own-box
    # gates, a segment's result taken where its hit lies within reach or the ray runs clear of the field, else walked again on the tree as
    # handed over; primary rays start there, as the library decides for this camera); the unit form with bare gates is the opt-in,
    # empirical one (include/vecchio_amd.h; VK_GATE_PROOF=0 prefers it)
    if form == "empirical":
        hs.desc.contents.flags = ffi.VK_SCENE_EMPIRICAL_TREES
        monkeypatch.setenv("VK_GATE_PROOF", "0")
    elif form == "near":
        # (the world has a GRID too — late round 5 — which the device walks only where the scene fits LDS; a larger scene is walked on
        # the near form's TREE, with the near form's conditions: EMU_GRID=0 is that view.  The device once dropped them with the grid it
        # did not use — 34 pixels of a grazing view of this scene; the second variant is that view's: a camera close to the ground
        # sphere, primary rays on the rebuilt tree)
        hs.desc.contents.flags = 0
        monkeypatch.setenv("EMU_GRID", "0")
        monkeypatch.setenv("EMU_PRIMARY_REF", "0" if variant.endswith("no-primary-ref") else "1")
    else:
        hs.desc.contents.flags = 0          # the grid form (the emulator's default for an eligible world)
    emu_ffi.take_redo_stats()
    img_x, ps_x, steps_x, info = emu.render_samples(hs.desc, cam, p)
    redone, segments = emu_ffi.take_redo_stats()
    compare(ps_o, ps_x, img_o, img_x)
    hs.desc.contents.flags = ffi.VK_SCENE_REFERENCE_TREE
    _, ps_r, steps_r, _ = emu.render_samples(hs.desc, cam, p)
    assert emu_ffi.take_redo_stats()[0] == 0
    assert np.array_equal(ps_x.view(np.uint32), ps_r.view(np.uint32))
    assert redone > 0, "nothing took the second walk: the scene does not exercise it"
    print(f"stress_spheres:{grid_half} {variant}: {redone} of {segments} segments / {ps_o.shape[0]} samples again; steps {steps_x} against {steps_r}")
    if variant == "global" and form != "grid" and grid_half > 100:      # (the small grid is mostly ground sphere, radius 1e5: every fifth segment is early)
        assert redone < 0.1 * segments and steps_x < 0.9 * steps_r


class SphereCrowd(Gen):
    """Worlds of spheres only, solid materials — what exact re-treeing rebuilds (vk_linearize.cpp rt_collect) — in the shapes that make
    computed hits precede their box entries: pixel-sized spheres far from the camera, a ground sphere of radius 1e3..1e5, grazing
    views; plus coincident copies (every hit an exact tie), hollow spheres (inverted boxes), objects beside subtrees and `len == 1`
    nodes in random tree shapes, sphere lights for the PDF integrator."""
    def __init__(self, seed, nasty=False):
        self.nasty = nasty
        self.r = np.random.default_rng(seed)
        self.d = Desc()                                      # (no image / checker / noise texture anywhere: the scene must stay sphere-only)
        self.lights = []
        self.use_pdf = bool(self.r.integers(0, 2))
        self.emit = self.d.light(6, 6, 6)
        self.surface = [self.d.lambertian(*self.r.uniform(0.2, 0.9, 3)), self.d.lambertian(*self.r.uniform(0.2, 0.9, 3)),
                        self.d.mat(ffi.VK_MAT_METAL, self.d.solid(0.8, 0.7, 0.6), float(self.r.uniform(0.0, 0.6))),
                        self.d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5)]

    def one(self, c, rad, m=None):
        c = np.asarray(c, np.float32); rad = np.float32(rad)
        ref = self.d.sphere(tuple(float(x) for x in c), float(rad), self.mat() if m is None else m)
        return ref, (c - rad, c + rad)                       # Sphere::bounding_box as the reference computes it (inverted when rad < 0)

    def build(self, far=None):
        r = self.r
        far = float(r.choice([1.0, 8.0, 40.0])) if far is None else far
        hollow = r.uniform() < 0.25
        objs = []
        for _ in range(int(r.integers(20, 160))):
            k = r.uniform()
            if k < 0.5:
                objs.append(self.one(self.pos(), r.uniform(0.1, 1.0)))
            elif k < 0.9:                                    # small spheres spread far out: seen from hundreds of radii away
                objs.append(self.one(r.uniform(-20, 30, 3) * far / 8.0 + 5.0, r.uniform(0.02, 0.3)))
            elif k < 0.95 and hollow:
                objs.append(self.one(self.pos(), -r.uniform(0.2, 0.8)))          # hollow (a scene with one keeps the tree handed over)
            else:
                c, rad = self.pos(), r.uniform(0.2, 0.9)                             # two materials on the same sphere: exact ties
                objs.append(self.one(c, rad)); objs.append(self.one(c, rad))
        if self.nasty:
            # overlapping clusters (final_scene's 1 000-sphere box), shells inside shells, concentric spheres, specks, a sphere around
            # everything (the camera inside glass)
            c0 = self.pos()
            for _ in range(int(r.integers(8, 60))):
                objs.append(self.one(c0 + r.uniform(-1.5, 1.5, 3), r.uniform(0.4, 1.4)))
            c1 = self.pos()
            for k in range(int(r.integers(2, 6))):
                objs.append(self.one(c1, 0.3 + 0.25 * k, m=self.surface[3]))
            for _ in range(int(r.integers(0, 12))):
                objs.append(self.one(self.pos(), float(r.choice([1e-6, 1e-4, 1e-3]))))
            if r.uniform() < 0.3:
                objs.append(self.one((5.0, 5.0, 5.0), float(r.choice([40.0, 400.0, 4000.0])), m=self.surface[3]))
        if r.uniform() < 0.6:
            R = float(r.choice([1.0e3, 1.0e4, 1.0e5]))
            objs.append(self.one((5.0, -R, 5.0), R))                                # the ground
        if self.use_pdf:
            lref, lbb = self.one(self.pos() + np.array([0, 6, 0]), r.uniform(0.5, 1.5), m=self.emit)
            objs.append((lref, lbb)); self.lights.append(lref)
        order = r.permutation(len(objs))
        world, _ = self.tree([objs[i] for i in order])
        desc = self.d.finish(world, self.lights if self.use_pdf else [])
        ang = r.uniform(0, 6.28)
        dist = float(r.choice([14.0, 60.0, 250.0])) * max(1.0, far / 8.0)
        cam = camera((5 + dist * np.cos(ang), r.uniform(0.3, 0.6) * dist, 5 + dist * np.sin(ang)), (5, 3, 5), vfov=float(r.choice([8.0, 25.0, 50.0])))
        kw = {} if self.use_pdf else dict(integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)
        return desc, cam, params(32, 24, 3, max_depth=int(r.choice([3, 12, 50])), seed=int(r.integers(1, 1000)), **kw)


class LayerWorld(SphereCrowd):
    """Worlds the GRID form applies to (vk_linearize.cpp rt_build_grid): hundreds of small spheres in a layer across y — on a jittered
    grid, in rows, in a ring, or scattered; layers far from the coordinate origin (rounding of the cell arithmetic), very thin and four
    times thicker ones, radii mixed within the 2.5-median-radii class — plus a ground sphere and a few large ones, random tree shapes as
    in SphereCrowd; cameras inside the layer, above it, far from it, grazing it."""
    def build(self, far=None):
        r = self.r
        n_side = int(r.choice([9, 14, 24, 40]))
        pitch = float(r.choice([0.5, 1.0, 2.5]))
        rad = float(r.choice([0.05, 0.2, 0.45])) * pitch
        centre = np.array([0.0, 0.0, 0.0]) if r.uniform() < 0.5 else r.uniform(-1.0, 1.0, 3) * np.array([3000.0, 40.0, 3000.0])
        thick = float(r.choice([0.0, 0.0, 1.5])) * rad
        shape = int(r.integers(0, 4))
        objs = []
        for i in range(n_side):
            for j in range(n_side):
                if shape == 1 and j % 3:                      # rows
                    continue
                if shape == 2 and not (0.25 * n_side ** 2 <= (i - n_side / 2) ** 2 + (j - n_side / 2) ** 2 <= 0.3 * n_side ** 2 + n_side):      # a ring
                    continue
                x, z = (i - n_side / 2 + r.uniform(0, 0.8)) * pitch, (j - n_side / 2 + r.uniform(0, 0.8)) * pitch
                if shape == 3:
                    x, z = r.uniform(-n_side / 2, n_side / 2, 2) * pitch
                rr = rad * float(r.choice([1.0, 1.0, 0.6, 1.8]))
                objs.append(self.one(centre + np.array([x, rr + r.uniform(0, 1) * thick, z]), rr))
        if len(objs) < 90:
            for _ in range(90 - len(objs)):
                x, z = r.uniform(-n_side / 2, n_side / 2, 2) * pitch
                objs.append(self.one(centre + np.array([x, rad, z]), rad))
        R = float(r.choice([1.0e3, 1.0e4]))
        objs.append(self.one(centre + np.array([0.0, -R, 0.0]), R))                 # the ground
        for _ in range(int(r.integers(0, 4))):                                       # large spheres standing in the layer
            big = rad * float(r.uniform(4.0, 9.0))
            objs.append(self.one(centre + np.array([r.uniform(-3, 3) * pitch, big, r.uniform(-3, 3) * pitch]), big, m=self.surface[int(r.integers(0, 4))]))
        order = r.permutation(len(objs))
        world, _ = self.tree([objs[i] for i in order])
        desc = self.d.finish(world, [])
        ext = n_side * pitch
        view = int(r.integers(0, 4))
        ang = r.uniform(0, 6.28)
        if view == 0:      # inside the layer, a sphere's height above the ground
            lf = centre + np.array([0.3 * ext * np.cos(ang), 2.5 * rad, 0.3 * ext * np.sin(ang)]); la = centre + np.array([0.0, rad, 0.0])
        elif view == 1:    # above it
            lf = centre + np.array([0.2 * ext, 0.8 * ext, 0.1 * ext]); la = centre
        elif view == 2:    # far away
            lf = centre + np.array([40.0 * ext * np.cos(ang), 9.0 * ext, 40.0 * ext * np.sin(ang)]); la = centre
        else:              # grazing
            lf = centre + np.array([2.0 * ext * np.cos(ang), 3.0 * rad, 2.0 * ext * np.sin(ang)]); la = centre + np.array([0.0, rad, 0.0])
        cam = camera(tuple(float(x) for x in lf), tuple(float(x) for x in la), vfov=float(r.choice([3.0, 25.0, 60.0])) if view != 2 else 2.0)
        return desc, cam, params(32, 24, 3, max_depth=int(r.choice([3, 12, 50])), seed=int(r.integers(1, 1000)),
                                 integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)


@pytest.mark.parametrize("seed", range(40))
def test_grid_form_on_layer_worlds_is_the_handed_over_tree_per_sample(seed, oracle, emu, built, monkeypatch):
    """the GRID form on forty random layer worlds: every sample the oracle's (equal draw counts, |dRGB|) and the handed-over tree's bit
    for bit; the grid walk's closest hit the closest hit over ALL spheres for 20 000 adversarial rays (tests/emu emu_grid_claims); and
    the same world on the near / unit form's TREE (EMU_GRID=0), which a scene too large for LDS is walked on"""
    import ctypes as C
    import emu_ffi
    desc, cam, p = LayerWorld(9000 + seed).build()
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    desc.contents.flags = ffi.VK_SCENE_REFERENCE_TREE
    img_r, ps_r, steps_r, info_r = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_r, img_o, img_r)
    desc.contents.flags = 0
    lib = emu_ffi.load()
    lib.emu_grid_claims.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    cnt = (C.c_uint64 * 3)(); v = (C.c_float * 8)()
    monkeypatch.delenv("EMU_GRID", raising=False)
    assert lib.emu_grid_claims(desc, 20000, seed + 1, cnt, v) == 0, "not a world the grid form applies to"
    assert cnt[2] == 0, f"{cnt[2]} of {cnt[0]} rays, e.g. o {list(v[0:3])} d {list(v[3:6])} grid {v[6]} all spheres {v[7]}"
    for env in ({}, {"EMU_GRID": "0"}, {"EMU_GRID": "0", "EMU_GLOBAL_VARIANT": "1"}):
        for k in ("EMU_GRID", "EMU_GLOBAL_VARIANT"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        img_x, ps_x, steps_x, info_x = emu.render_samples(desc, cam, p)
        assert np.array_equal(ps_x.view(np.uint32), ps_r.view(np.uint32)), (env, int((ps_x.view(np.uint32) != ps_r.view(np.uint32)).any(axis=1).sum()))


@pytest.mark.parametrize("seed", range(60))
def test_exact_retree_on_sphere_crowds_is_the_handed_over_tree_per_sample(seed, oracle, emu, built, monkeypatch):
    import emu_ffi
    desc, cam, p = SphereCrowd(7000 + seed, nasty=seed >= 40).build()
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    desc.contents.flags = ffi.VK_SCENE_REFERENCE_TREE
    img_r, ps_r, steps_r, info_r = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_r, img_o, img_r)
    # the default (the near form where its reach spans the world, else the unit form with grown gates where it is cheap, else the near
    # form whatever its reach, else the tree as handed over), the unit form wherever it is eligible (VK_NEAR_FIRST=0), the near form
    # wherever it is (VK_UNIT_FORM=0: with and without primary rays starting on the tree as handed over), and the opt-in empirical form
    for flags, env in ((0, {}), (0, {"VK_NEAR_FIRST": "0"}), (0, {"VK_UNIT_FORM": "0"}), (0, {"VK_UNIT_FORM": "0", "EMU_PRIMARY_REF": "1"}),
                       (ffi.VK_SCENE_EMPIRICAL_TREES, {"VK_GATE_PROOF": "0"})):
        desc.contents.flags = flags
        for k in ("VK_UNIT_FORM", "VK_NEAR_FIRST", "EMU_PRIMARY_REF", "VK_GATE_PROOF"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for variant in ("0", "1"):                           # the device's LDS form (whole samples again) and its global-memory form
            monkeypatch.setenv("EMU_GLOBAL_VARIANT", variant)
            img_x, ps_x, steps_x, info_x = emu.render_samples(desc, cam, p)
            assert np.array_equal(ps_x.view(np.uint32), ps_r.view(np.uint32)), (flags, env, variant,
                int((ps_x.view(np.uint32) != ps_r.view(np.uint32)).any(axis=1).sum()))


def test_a_tree_whose_boxes_do_not_hold_their_spheres_is_walked_as_handed_over(oracle, emu, built):
    """The tree is the caller's: a node box that cuts into one of its spheres (which no BVHNode::new tree has) changes which hits the
    reference finds.  Exact re-treeing assumes a sphere's own box inside its unit's, so such a tree must be left alone — same items,
    same samples as with VK_SCENE_REFERENCE_TREE, and the oracle's."""
    g = SphereCrowd(7001)                                   # (a world that IS rebuilt when its boxes are right)
    desc, cam, p = g.build()
    d = desc.contents
    cut = 0
    for i in range(d.n_bvh):                                 # shrink the box of every node that holds a sphere directly
        n = d.bvh[i]
        if (n.left >> 28) == ffi.VK_KIND_SPHERE and n.bb_max[0] - n.bb_min[0] > 0.05:
            n.bb_max[0] -= 0.02; cut += 1
    assert cut > 10
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    desc.contents.flags = ffi.VK_SCENE_REFERENCE_TREE
    img_r, ps_r, steps_r, info_r = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_r, img_o, img_r)
    for flags in (0, ffi.VK_SCENE_EMPIRICAL_TREES):
        desc.contents.flags = flags
        img_x, ps_x, steps_x, info_x = emu.render_samples(desc, cam, p)
        assert steps_x == steps_r and np.array_equal(ps_x.view(np.uint32), ps_r.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_grid_form_on_layer_worlds_on_the_gpu(seed, device, oracle):
    """the GRID form through the C ABI on random layer worlds (LayerWorld above): every sample the oracle's, and the handed-over tree's
    bit for bit"""
    from test_gpu_parity import compare_samples, device_samples
    from vecchio_amd import DeviceScene
    desc, cam, p = LayerWorld(9100 + seed).build()
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    out = []
    for flags in (0, ffi.VK_SCENE_REFERENCE_TREE):
        desc.contents.flags = flags
        ds = DeviceScene(desc)
        if flags == 0:      # (the grid form where the scene fits LDS — most of these — else the near form's tree, exact too)
            assert ds.info().tree in (ffi.VK_TREE_REBUILT_GRID, ffi.VK_TREE_REBUILT_NEAR, ffi.VK_TREE_REBUILT_PROVEN), ds.info().tree
            assert ds.info().tree == ffi.VK_TREE_REBUILT_GRID or not ds.info().lds_bytes
        img_d, ps_d = device_samples(ds, cam, p)
        compare_samples(ps_o, ps_d, img_o, img_d)
        out.append(ps_d)
        ds.close()
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_exact_retree_on_sphere_crowds_on_the_gpu(seed, device, oracle):
    from test_gpu_parity import compare_samples, device_samples
    from vecchio_amd import DeviceScene
    desc, cam, p = SphereCrowd(7100 + seed).build()
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    imgs = []
    for flags in (0, ffi.VK_SCENE_EMPIRICAL_TREES, ffi.VK_SCENE_REFERENCE_TREE):
        desc.contents.flags = flags
        ds = DeviceScene(desc)
        img_d, ps_d = device_samples(ds, cam, p)
        compare_samples(ps_o, ps_d, img_o, img_d)
        imgs.append(ps_d)
        ds.close()
    assert np.array_equal(imgs[0].view(np.uint32), imgs[2].view(np.uint32))
    assert np.array_equal(imgs[1].view(np.uint32), imgs[2].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("switch,tree", [("VK_UNIT_FORM", ffi.VK_TREE_REBUILT_NEAR)])
def test_a_form_forced_on_sphere_crowds_on_the_gpu(switch, tree, device, oracle):
    """the near form (vk_linearize.cpp rt_grow_near) wherever it is eligible (VK_UNIT_FORM=0) — the default takes it first only where
    its reach spans the world.  (The unit form is never cheap on these crowds — their leaf units are long — so forcing IT has its own
    test on the InOneWeekend scene: test_gpu_exact_retree.py test_unit_form_forced_on_the_gpu.)  The switches are honoured by the
    DEBUG build of the library only, and read at scene creation, so this runs in a child process: twelve crowds (four of them with
    coincident spheres), every sample the oracle's and the handed-over tree's"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "import oracle_ffi as O\n"
            "from test_retree import SphereCrowd\n"
            "from test_gpu_parity import compare_samples, device_samples\n"
            "from vecchio_amd import DeviceScene, ffi\n"
            "dbg = ffi.load_debug_lib(); trees = []\n"
            "for seed in list(range(8)) + [40, 41, 42, 43]:\n"
            "    desc, cam, p = SphereCrowd(7200 + seed, nasty=seed >= 40).build()\n"
            "    img_o, ps_o = O.render_samples(desc, cam, p)\n"
            "    out = []\n"
            "    for flags in (0, ffi.VK_SCENE_REFERENCE_TREE):\n"
            "        desc.contents.flags = flags\n"
            "        ds = DeviceScene(desc, lib=dbg); trees.append(ds.info().tree)\n"
            "        img_d, ps_d = device_samples(ds, cam, p); compare_samples(ps_o, ps_d, img_o, img_d); out.append(ps_d); ds.close()\n"
            "    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32)), seed\n"
            "print('TREES', sorted(set(trees)))\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{switch: "0"}), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TREES" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert str(tree) in r.stdout.split("TREES")[1], r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_retreed_crowd_on_the_gpu(seed, device, oracle):
    import ctypes as C
    from test_gpu_parity import compare_samples, device_samples
    from vecchio_amd import DeviceScene
    desc, cam, p = Crowd(5100 + seed).build()
    ds = DeviceScene(desc)
    img_d, ps_d = device_samples(ds, cam, p)
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    compare_samples(ps_o, ps_d, img_o, img_d)
    ds.close()


def test_hollow_sphere_keeps_the_reference_tree(oracle, emu, built):
    """A negative-radius sphere (the hollow glass sphere of scene.rs:123-127) has an INVERTED bounding box in the reference
    (center -/+ radius, hittable.rs:97-102), which shrinks its ancestors' boxes: hits on it depend on the tree handed over.
    With VK_SCENE_FAST_ACCEL the lineariser must therefore leave a subtree that holds one alone — 24 spheres, one hollow,
    boxes computed as BVHNode::new's surrounding_box would (accel.rs:117-131)."""
    g = Crowd(77)
    g.use_pdf = False
    objs = []
    for i in range(24):
        c = g.pos().astype(np.float32)
        rad = np.float32(-0.9 if i == 11 else g.r.uniform(0.3, 1.0))
        ref = g.d.sphere(tuple(c), float(rad), g.surface[4] if i == 11 else g.mat())
        objs.append((ref, (c - rad, c + rad)))                      # inverted for the hollow one, as the reference computes it
    world, _ = g.tree(objs)
    desc = g.d.finish(world, [])
    cam = camera((18, 6, 14), (5, 5, 5), vfov=50.0)
    p = params(24, 20, 4, max_depth=12, seed=5, integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    img_r, ps_r, steps_r, info_r = emu.render_samples(desc, cam, p)            # flags = 0
    compare(ps_o, ps_r, img_o, img_r)
    desc.contents.flags = ffi.VK_SCENE_FAST_ACCEL
    img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    assert (info[0], steps) == (info_r[0], steps_r)                           # not rebuilt: the very same walk
