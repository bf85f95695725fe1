"""Randomised scene graphs: every seed builds a small scene out of everything the boundary accepts — spheres
(also negative radius), moving spheres, rects, Boxys, generic lists, media with sphere / Boxy boundaries,
Translate / Rotate chains around primitives, lists, media and nested BVHs, FlipFace bits, random BVH shapes with
real bounding boxes (so culling, skip links and always-hit leaves are exercised), random light lists and material
mixes — and the kernel's formulation (tests/emu) must agree with the recursive oracle on every sample."""
import numpy as np
import pytest

from descs import Desc, camera, params
from test_emu_parity import compare
from vecchio_amd import ffi


def _union(a, b):
    return (np.minimum(a[0], b[0]), np.maximum(a[1], b[1]))


def _corners(bb):
    lo, hi = bb
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], np.float64)


class Gen:
    def __init__(self, seed):
        self.r = np.random.default_rng(seed)
        self.d = Desc()
        self.lights = []
        self.use_pdf = bool(self.r.integers(0, 2))
        tex = self.d.image(self.r.integers(0, 256, (8, 8, 3)))
        chk = self.d.checker(self.d.solid(0.1, 0.1, 0.1), tex)
        self.surface = [self.d.lambertian(*self.r.uniform(0.2, 0.9, 3)),
                        self.d.mat(ffi.VK_MAT_LAMBERTIAN, chk),
                        self.d.mat(ffi.VK_MAT_LAMBERTIAN, tex),
                        self.d.mat(ffi.VK_MAT_METAL, self.d.solid(0.8, 0.7, 0.6), float(self.r.uniform(0.0, 0.6))),
                        self.d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5)]
        if self.use_pdf:
            self.surface.append(self.d.mat(ffi.VK_MAT_SPEC_DIFFUSE, 0, 0.5, self.surface[3], self.surface[0]))
        self.iso = self.d.mat(ffi.VK_MAT_ISOTROPIC, self.d.solid(0.8, 0.8, 0.9))
        self.emit = self.d.light(6, 6, 6)

    def mat(self):
        return self.surface[int(self.r.integers(0, len(self.surface)))]

    def pos(self):
        return self.r.uniform(1.0, 9.0, 3)

    # each maker returns (ref, (lo, hi)) with a conservative box
    def sphere(self, m=None):
        c, rad = self.pos(), float(self.r.uniform(0.3, 1.2))
        if self.r.uniform() < 0.1:
            rad = -rad
        return self.d.sphere(tuple(c), rad, self.mat() if m is None else m), (c - abs(rad) - 1e-3, c + abs(rad) + 1e-3)

    def moving(self):
        c0 = self.pos(); c1 = c0 + self.r.uniform(-0.7, 0.7, 3); rad = float(self.r.uniform(0.3, 0.8))
        return (self.d.moving_sphere(tuple(c0), tuple(c1), 0.0, 1.0, rad, self.mat()),
                (np.minimum(c0, c1) - rad - 1e-3, np.maximum(c0, c1) + rad + 1e-3))

    def rect(self, m=None):
        ax = [(0, 1, 2), (0, 2, 1), (1, 2, 0)][int(self.r.integers(0, 3))]
        p, s = self.pos(), self.r.uniform(0.5, 2.0, 2)
        k = float(p[ax[2]])
        lo, hi = np.zeros(3), np.zeros(3)
        lo[ax[0]], hi[ax[0]] = p[ax[0]] - s[0], p[ax[0]] + s[0]
        lo[ax[1]], hi[ax[1]] = p[ax[1]] - s[1], p[ax[1]] + s[1]
        lo[ax[2]], hi[ax[2]] = k - 1e-3, k + 1e-3
        ref = self.d.rect(float(lo[ax[0]]), float(hi[ax[0]]), float(lo[ax[1]]), float(hi[ax[1]]), k, ax, self.mat() if m is None else m)
        return ref, (lo, hi)

    def boxy(self, m=None):
        p0 = self.pos(); p1 = p0 + self.r.uniform(0.4, 2.0, 3)
        return self.d.boxy(tuple(p0), tuple(p1), self.mat() if m is None else m), (p0 - 1e-3, p1 + 1e-3)

    def generic_list(self):
        items, bb = [], None
        for _ in range(int(self.r.integers(1, 4))):
            ref, b = [self.sphere, self.rect, self.moving][int(self.r.integers(0, 3))]()
            if self.r.uniform() < 0.3:
                ref = Desc.flip(ref)
            items.append(ref); bb = b if bb is None else _union(bb, b)
        return self.d.list_(items), bb

    def medium(self):
        ref, bb = (self.sphere if self.r.uniform() < 0.6 else self.boxy)(m=self.surface[4])
        return self.d.medium(ref, float(self.r.uniform(0.2, 2.0)), self.iso), bb

    def wrapped(self, depth=0):
        kind = int(self.r.integers(0, 6))
        if kind == 0:
            ref, bb = self.sphere()
        elif kind == 1:
            ref, bb = self.boxy()
        elif kind == 2:
            ref, bb = self.generic_list()
        elif kind == 3:
            ref, bb = self.medium()
        elif kind == 4:
            ref, bb = self.tree([self.sphere() for _ in range(int(self.r.integers(2, 5)))])
        else:
            ref, bb = self.rect()
        for _ in range(int(self.r.integers(1, 6 if depth == 0 else 3))):          # up to 5 wrappers: splits instance records
            if self.r.uniform() < 0.5:
                off = self.r.uniform(-1.5, 1.5, 3)
                ref, bb = self.d.translate(ref, tuple(off)), (bb[0] + off, bb[1] + off)
            else:
                axis, deg = int(self.r.integers(0, 3)), float(self.r.uniform(-40, 40))
                ref = self.d.rotate(ref, axis, deg)
                a = np.deg2rad(deg); c, s = np.cos(a), np.sin(a)
                P = _corners(bb)
                i, j = [(1, 2), (0, 2), (0, 1)][axis]
                Q = P.copy()
                # bound of both rotation senses (the box only has to contain the object)
                Q1, Q2 = P.copy(), P.copy()
                Q1[:, i], Q1[:, j] = c * P[:, i] - s * P[:, j], s * P[:, i] + c * P[:, j]
                Q2[:, i], Q2[:, j] = c * P[:, i] + s * P[:, j], -s * P[:, i] + c * P[:, j]
                Q = np.concatenate([Q1, Q2])
                bb = (Q.min(0) - 1e-2, Q.max(0) + 1e-2)
            if self.r.uniform() < 0.15:
                ref = Desc.flip(ref)
        return ref, bb

    def tree(self, objs):
        """random binary tree over (ref, box) pairs; BVH children and bare objects mixed like BVHNode::new never does"""
        if len(objs) == 1:
            ref, bb = objs[0]
            if self.r.uniform() < 0.5:
                return self.d.bvh_node(ref, ref, tuple(bb[0]), tuple(bb[1])), bb       # len == 1: both children the object
            return ref, bb
        k = int(self.r.integers(1, len(objs)))
        (l, lb), (r_, rb) = self.tree(objs[:k]), self.tree(objs[k:])
        bb = _union(lb, rb)
        return self.d.bvh_node(l, r_, tuple(bb[0].astype(np.float32)), tuple(bb[1].astype(np.float32))), bb

    def build(self):
        objs = []
        for _ in range(int(self.r.integers(3, 9))):
            k = self.r.uniform()
            objs.append(self.sphere() if k < 0.25 else self.moving() if k < 0.35 else self.rect() if k < 0.5 else
                        self.boxy() if k < 0.6 else self.generic_list() if k < 0.68 else self.medium() if k < 0.78 else self.wrapped())
        # lights: an emitting rect (flipped in the world like every scene builder does), sometimes a sphere / Boxy light too
        lref, lbb = self.rect(m=self.emit)
        objs.append((Desc.flip(lref), lbb)); self.lights.append(lref)
        if self.r.uniform() < 0.4:
            sref, sbb = self.sphere(m=self.emit); objs.append((sref, sbb)); self.lights.append(sref)
        if self.r.uniform() < 0.3:
            bref, bbb = self.boxy(m=self.emit); objs.append((bref, bbb)); self.lights.append(bref)
        order = self.r.permutation(len(objs))
        world, _ = self.tree([objs[i] for i in order])
        if (world >> 28) != ffi.VK_KIND_BVH:
            world = self.d.big_box(world, world)
        desc = self.d.finish(world, self.lights if self.use_pdf else [])
        ang = self.r.uniform(0, 6.28)
        cam = camera((5 + 11 * np.cos(ang), self.r.uniform(3, 8), 5 + 11 * np.sin(ang)), (5, 5, 5), vfov=50.0,
                     aperture=float(self.r.choice([0.0, 0.3])))
        if self.use_pdf:
            p = params(20, 16, 4, max_depth=int(self.r.choice([2, 8, 50])), seed=int(self.r.integers(1, 1000)))
        else:
            p = params(20, 16, 4, max_depth=int(self.r.choice([2, 8, 50])), seed=int(self.r.integers(1, 1000)),
                       integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)
        return desc, cam, p


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_graph(seed, oracle, emu, built):
    g = Gen(1000 + seed)
    desc, cam, p = g.build()
    img_o, ps_o = oracle.render_samples(desc, cam, p)
    img_e, ps_e, steps, info = emu.render_samples(desc, cam, p)
    compare(ps_o, ps_e, img_o, img_e)
    assert steps > 0
