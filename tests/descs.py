"""Hand-built vk_scene_desc graphs for unit tests (tiny scenes with closed-form answers)."""
import ctypes as C

import numpy as np

from vecchio_amd import ffi


class Desc:
    """Collects records like the Rust shim's FlatBuilder and keeps the ctypes arrays alive."""

    def __init__(self):
        self.bvh, self.spheres, self.moving, self.rects, self.lists, self.list_items = [], [], [], [], [], []
        self.media, self.translates, self.rotates, self.materials, self.textures = [], [], [], [], []
        self.images, self.perlins, self.lights = [], [], []
        self._keep = []
        self.world = 0

    # textures / materials
    def solid(self, r, g, b):
        self.textures.append(ffi.Texture(ffi.VK_TEX_SOLID, ffi.F3(r, g, b), 0, 0, 0.0))
        return len(self.textures) - 1

    def checker(self, odd, even):
        self.textures.append(ffi.Texture(ffi.VK_TEX_CHECKER, ffi.F3(0, 0, 0), odd, even, 0.0))
        return len(self.textures) - 1

    def image(self, rgb8):
        arr = np.ascontiguousarray(rgb8, dtype=np.uint8)
        self._keep.append(arr)
        h, w, _ = arr.shape
        self.images.append(ffi.Image(w, h, arr.ctypes.data_as(C.POINTER(C.c_uint8))))
        self.textures.append(ffi.Texture(ffi.VK_TEX_IMAGE, ffi.F3(0, 0, 0), len(self.images) - 1, 0, 0.0))
        return len(self.textures) - 1

    def noise(self, scale, seed=0):
        """NoiseTexture (material.rs:416-434) over a seeded Perlin table (material.rs:355-377: 256 unit vectors, three permutations)"""
        rng = np.random.default_rng(1000 + seed)
        pl = ffi.Perlin()
        v = rng.uniform(-1, 1, (256, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        for k in range(256):
            for c in range(3):
                pl.ranvec[k][c] = float(np.float32(v[k, c]))
        for name in ("perm_x", "perm_y", "perm_z"):
            perm = rng.permutation(256)
            for k in range(256):
                getattr(pl, name)[k] = int(perm[k])
        self.perlins.append(pl)
        self.textures.append(ffi.Texture(ffi.VK_TEX_NOISE, ffi.F3(0, 0, 0), len(self.perlins) - 1, 0, float(scale)))
        return len(self.textures) - 1

    def mat(self, kind, tex=0, param=0.0, a=0, b=0):
        self.materials.append(ffi.Material(kind, tex, param, a, b))
        return len(self.materials) - 1

    def lambertian(self, r, g, b):
        return self.mat(ffi.VK_MAT_LAMBERTIAN, self.solid(r, g, b))

    def light(self, r, g, b):
        return self.mat(ffi.VK_MAT_DIFFUSE_LIGHT, self.solid(r, g, b))

    # hittables
    def sphere(self, c, r, m):
        self.spheres.append(ffi.Sphere(ffi.F3(*c), r, m))
        return ffi.make_ref(ffi.VK_KIND_SPHERE, len(self.spheres) - 1)

    def moving_sphere(self, c0, c1, t0, t1, r, m):
        self.moving.append(ffi.MovingSphere(ffi.F3(*c0), ffi.F3(*c1), t0, t1, r, m))
        return ffi.make_ref(ffi.VK_KIND_MOVING_SPHERE, len(self.moving) - 1)

    def rect(self, c0, c1, d0, d1, k, axes, m):
        self.rects.append(ffi.Rect(c0, c1, d0, d1, k, axes[0], axes[1], axes[2], 0, m))
        return ffi.make_ref(ffi.VK_KIND_RECT, len(self.rects) - 1)

    def xy_rect(self, x0, x1, y0, y1, k, m):
        return self.rect(x0, x1, y0, y1, k, (0, 1, 2), m)

    def xz_rect(self, x0, x1, z0, z1, k, m):
        return self.rect(x0, x1, z0, z1, k, (0, 2, 1), m)

    def yz_rect(self, y0, y1, z0, z1, k, m):
        return self.rect(y0, y1, z0, z1, k, (1, 2, 0), m)

    @staticmethod
    def flip(ref):
        return ref ^ ffi.VK_REF_FLIP

    def list_(self, refs):
        self.lists.append(ffi.List(len(self.list_items), len(refs)))
        self.list_items.extend(refs)
        return ffi.make_ref(ffi.VK_KIND_LIST, len(self.lists) - 1)

    def boxy(self, p0, p1, m):
        """Boxy::new, hittable.rs:321-359"""
        return self.list_([
            self.xy_rect(p0[0], p1[0], p0[1], p1[1], p1[2], m),
            self.flip(self.xy_rect(p0[0], p1[0], p0[1], p1[1], p0[2], m)),
            self.xz_rect(p0[0], p1[0], p0[2], p1[2], p1[1], m),
            self.flip(self.xz_rect(p0[0], p1[0], p0[2], p1[2], p0[1], m)),
            self.yz_rect(p0[1], p1[1], p0[2], p1[2], p1[0], m),
            self.flip(self.yz_rect(p0[1], p1[1], p0[2], p1[2], p0[0], m)),
        ])

    def medium(self, boundary, density, m_iso):
        self.media.append(ffi.Medium(boundary, -1.0 / density, m_iso))
        return ffi.make_ref(ffi.VK_KIND_MEDIUM, len(self.media) - 1)

    def translate(self, child, off):
        self.translates.append(ffi.Translate(child, ffi.F3(*off)))
        return ffi.make_ref(ffi.VK_KIND_TRANSLATE, len(self.translates) - 1)

    def rotate(self, child, axis, deg):
        rad = np.float32(deg) * np.float32(np.pi / 180.0)
        self.rotates.append(ffi.Rotate(child, axis, float(np.sin(rad, dtype=np.float32)), float(np.cos(rad, dtype=np.float32))))
        return ffi.make_ref(ffi.VK_KIND_ROTATE, len(self.rotates) - 1)

    def bvh_node(self, left, right, bmin, bmax):
        self.bvh.append(ffi.BvhNode(ffi.F3(*bmin), ffi.F3(*bmax), left, right))
        return ffi.make_ref(ffi.VK_KIND_BVH, len(self.bvh) - 1)

    def big_box(self, left, right):
        return self.bvh_node(left, right, (-1e6, -1e6, -1e6), (1e6, 1e6, 1e6))

    def finish(self, world, lights=()):
        def arr(T, items):
            a = (T * max(1, len(items)))(*items)
            self._keep.append(a)
            return a
        d = ffi.SceneDesc()
        d.abi_version = ffi.VK_ABI_VERSION
        d.n_bvh, d.bvh = len(self.bvh), arr(ffi.BvhNode, self.bvh)
        d.n_spheres, d.spheres = len(self.spheres), arr(ffi.Sphere, self.spheres)
        d.n_moving_spheres, d.moving_spheres = len(self.moving), arr(ffi.MovingSphere, self.moving)
        d.n_rects, d.rects = len(self.rects), arr(ffi.Rect, self.rects)
        d.n_lists, d.lists = len(self.lists), arr(ffi.List, self.lists)
        d.n_list_items, d.list_items = len(self.list_items), arr(C.c_uint32, self.list_items)
        d.n_media, d.media = len(self.media), arr(ffi.Medium, self.media)
        d.n_translates, d.translates = len(self.translates), arr(ffi.Translate, self.translates)
        d.n_rotates, d.rotates = len(self.rotates), arr(ffi.Rotate, self.rotates)
        d.n_materials, d.materials = len(self.materials), arr(ffi.Material, self.materials)
        d.n_textures, d.textures = len(self.textures), arr(ffi.Texture, self.textures)
        d.n_images, d.images = len(self.images), arr(ffi.Image, self.images)
        d.n_perlins, d.perlins = len(self.perlins), arr(ffi.Perlin, self.perlins)
        d.world = world
        d.n_lights, d.lights = len(lights), arr(C.c_uint32, list(lights))
        self.desc = d
        return C.pointer(d)


def camera(lookfrom, lookat, vfov=40.0, aspect=1.0, aperture=0.0, focus=10.0, t0=0.0, t1=1.0, vup=(0, 1, 0)):
    lib = ffi.load_host_lib()
    cam = ffi.Camera()
    lib.vkh_camera_new(ffi.F3(*lookfrom), ffi.F3(*lookat), ffi.F3(*vup), vfov, aspect, aperture, focus, t0, t1, C.byref(cam))
    return cam


def params(width, height, spp, max_depth=50, seed=2, integrator=ffi.VK_INTEGRATOR_PDF, background=ffi.VK_BACKGROUND_SOLID,
           bg=(0.0, 0.0, 0.0), tile_rank=0, tile_world=1):
    p = ffi.RenderParams()
    p.width, p.height, p.samples_per_pixel, p.max_depth, p.seed = width, height, spp, max_depth, seed
    p.integrator, p.background = integrator, background
    p.background_color = ffi.F3(*bg)
    p.tile_rank, p.tile_world = tile_rank, tile_world
    return p
