"""Small hand-built scenes that exercise the corners of the scene graph the reference's own
builders do not (or only barely) reach: nested transform chains, mixed BVH children, flipped
subtrees, a Boxy light, a sphere light, SpecDiffuse, box-bounded media, image-textured media."""
import numpy as np

from descs import Desc, camera, params
from vecchio_amd import ffi


def _room(d, light_ref_list):
    """a closed grey box room with an XZ light near the ceiling; returns list of world refs"""
    white = d.lambertian(0.73, 0.73, 0.73)
    red = d.lambertian(0.65, 0.05, 0.05)
    refs = [
        Desc.flip(d.yz_rect(0, 10, 0, 10, 10, red)), d.yz_rect(0, 10, 0, 10, 0, white),
        Desc.flip(d.xz_rect(0, 10, 0, 10, 0, white)), d.xz_rect(0, 10, 0, 10, 10, white),
        Desc.flip(d.xy_rect(0, 10, 0, 10, 10, white)),
    ]
    lm = d.light(12, 12, 12)
    ls = d.xz_rect(3, 7, 3, 7, 9.9, lm)
    refs.append(Desc.flip(ls))
    light_ref_list.append(ls)
    return refs


def _bvh_chain(d, refs):
    """left-deep BVH with huge boxes: exercises traversal order/tie rules, not culling"""
    node = d.big_box(refs[0], refs[1]) if len(refs) > 1 else d.big_box(refs[0], refs[0])
    for r in refs[2:]:
        node = d.big_box(node, r)          # BVH child on the left, object child on the right (mixed)
    return node


def nested_transforms():
    d = Desc()
    lights = []
    refs = _room(d, lights)
    glass = d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5)
    metal = d.mat(ffi.VK_MAT_METAL, d.solid(0.8, 0.8, 0.9), 0.3)
    box = d.boxy((0, 0, 0), (2, 3, 2), d.lambertian(0.2, 0.6, 0.3))
    # 5 wrappers -> splits into nested instance records (MAX_OPS = 4)
    chain = d.translate(d.rotate(d.rotate(d.rotate(d.translate(box, (0.5, 0, 0.5)), 0, 10.0), 2, -7.0), 1, 25.0), (3, 0, 4))
    refs.append(chain)
    # an instanced BVH of spheres, itself containing an instance
    inner = d.big_box(d.sphere((0, 0, 0), 0.7, glass), d.translate(d.sphere((0, 0, 0), 0.5, metal), (0, 1.4, 0)))
    refs.append(d.translate(d.rotate(inner, 1, 40.0), (7, 1.0, 6)))
    refs.append(Desc.flip(d.translate(d.sphere((0, 0, 0), 0.6, d.lambertian(0.9, 0.5, 0.1)), (2, 0.6, 8))))
    # a Vec that is NOT the Boxy pattern (generic list path: first-wins ties, mixed primitive kinds)
    grey = d.lambertian(0.5, 0.5, 0.5)
    refs.append(d.list_([d.sphere((8, 3, 2), 0.8, grey), d.xy_rect(7, 9, 2, 4, 2.0, metal), Desc.flip(d.xy_rect(7, 9, 2, 4, 2.0, grey)),
                         d.moving_sphere((8, 5, 2), (8.5, 5, 2), 0.0, 1.0, 0.5, grey)]))
    world = _bvh_chain(d, refs)
    desc = d.finish(world, lights)
    cam = camera((5, 5, -12), (5, 5, 0), vfov=40.0)
    return d, desc, cam, params(48, 48, 8)


def lights_and_specdiffuse():
    d = Desc()
    lights = []
    refs = _room(d, lights)
    # a sphere light and a Boxy light in the lights list (hittable.rs:104-134, 371-377)
    sl = d.sphere((2, 8, 2), 0.5, d.light(20, 18, 15))
    refs.append(sl)
    lights.append(sl)
    bl = d.boxy((7, 7, 7), (8, 8, 8), d.light(10, 10, 14))
    refs.append(bl)
    lights.append(bl)
    lights.append(Desc.flip(lights[0]))     # a FlipFace in lights: trait defaults (pdf 0, direction (1,0,0))
    spec = d.mat(ffi.VK_MAT_METAL, d.solid(0.9, 0.9, 0.9), 0.05)
    diff = d.lambertian(0.3, 0.3, 0.8)
    sd = d.mat(ffi.VK_MAT_SPEC_DIFFUSE, 0, 0.4, spec, diff)
    refs.append(d.sphere((5, 2, 5), 2.0, sd))
    world = _bvh_chain(d, refs)
    desc = d.finish(world, lights)
    cam = camera((5, 5, -12), (5, 5, 0), vfov=40.0)
    return d, desc, cam, params(48, 48, 8)


def media_and_textures():
    d = Desc()
    lights = []
    refs = _room(d, lights)
    rng = np.random.default_rng(5)
    tex = d.image(rng.integers(0, 256, (16, 32, 3)))
    iso_img = d.mat(ffi.VK_MAT_ISOTROPIC, tex)
    iso = d.mat(ffi.VK_MAT_ISOTROPIC, d.solid(0.9, 0.9, 0.9))
    glass = d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5)
    b1 = d.sphere((3, 3, 5), 2.0, glass)
    refs.append(d.medium(b1, 0.6, iso_img))                      # image-textured phase function: needs rec1.u/v
    bx = d.boxy((6, 0, 2), (9, 4, 6), d.lambertian(1, 1, 1))
    refs.append(d.medium(bx, 0.4, iso))                          # Boxy boundary
    refs.append(d.translate(d.medium(d.sphere((0, 0, 0), 1.0, glass), 1.5, iso), (5, 7, 7)))   # medium under a transform
    chk = d.checker(d.solid(0.1, 0.1, 0.1), tex)
    refs.append(d.sphere((5, 1.0, 2), 1.0, d.mat(ffi.VK_MAT_LAMBERTIAN, chk)))
    refs.append(d.moving_sphere((1, 6, 8), (2, 6, 8), 0.0, 1.0, 0.8, d.lambertian(0.7, 0.3, 0.1)))
    # a single-object BVH node holding a medium: tested (and drawn) twice (accel.rs:102-107)
    single = d.medium(d.sphere((8, 8, 8), 1.0, glass), 2.0, iso)
    refs.append(d.big_box(single, single))
    world = _bvh_chain(d, refs)
    desc = d.finish(world, lights)
    cam = camera((5, 5, -12), (5, 5, 0), vfov=40.0)
    return d, desc, cam, params(48, 48, 8)


def scatter_sky():
    """IOW-style: scatter integrator + sky, metal/dielectric/negative radius, defocus blur"""
    d = Desc()
    refs = [d.sphere((0, -100.5, -1), 100.0, d.lambertian(0.8, 0.8, 0.0)),
            d.sphere((0, 0, -1), 0.5, d.lambertian(0.1, 0.2, 0.5)),
            d.sphere((1, 0, -1), 0.5, d.mat(ffi.VK_MAT_METAL, d.solid(0.8, 0.6, 0.2), 0.3)),
            d.sphere((-1, 0, -1), 0.5, d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5)),
            d.sphere((-1, 0, -1), -0.45, d.mat(ffi.VK_MAT_DIELECTRIC, 0, 1.5)),
            d.sphere((0, 1.2, -1), 0.3, d.light(4, 4, 4))]
    world = _bvh_chain(d, refs)
    desc = d.finish(world, [])
    cam = camera((3, 3, 2), (0, 0, -1), vfov=20.0, aspect=1.5, aperture=0.5, focus=5.2)
    return d, desc, cam, params(48, 32, 8, integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)


def metal_only():
    """spheres of fuzzy Metal only, filling the frame: every shading lane of a wave wants a random_in_unit_sphere point at once — the
    device deals ONE candidate per lane and round (vk_kernels.h cooperative_ball, K = 64, G = 1) and regroups until the slowest
    lane has accepted; paths bounce between the mirrors to the depth limit"""
    d = Desc()
    refs = [d.sphere((0, -1000.0, 0), 1000.0, d.mat(ffi.VK_MAT_METAL, d.solid(0.9, 0.9, 0.9), 0.6))]
    for i in range(-3, 4):
        for j in range(-3, 4):
            refs.append(d.sphere((1.1 * i, 0.5, 1.1 * j), 0.5, d.mat(ffi.VK_MAT_METAL, d.solid(0.6 + 0.05 * i, 0.7, 0.6 + 0.05 * j),
                                                                     0.05 * ((i + j) % 7 + 3))))
    world = _bvh_chain(d, refs)
    desc = d.finish(world, [])
    cam = camera((4, 6, 5), (0, 0, 0), vfov=35.0, aspect=1.0, aperture=0.0, focus=8.0)
    return d, desc, cam, params(64, 64, 8, integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)


def one_metal_among_many():
    """one small fuzzy Metal sphere among Lambertian ones: a wave seldom has more than a lane or two that want a sphere point — the
    cooperative sampler's other extreme (K = 1: all 64 lanes draw candidates for one owner), and most phases have none at all"""
    d = Desc()
    refs = [d.sphere((0, -1000.0, 0), 1000.0, d.lambertian(0.5, 0.5, 0.5))]
    for i in range(-3, 4):
        for j in range(-3, 4):
            m = d.mat(ffi.VK_MAT_METAL, d.solid(0.9, 0.8, 0.7), 1.0) if (i, j) == (0, 0) else d.lambertian(0.3 + 0.1 * (i % 3), 0.5, 0.4)
            refs.append(d.sphere((1.1 * i, 0.5, 1.1 * j), 0.5, m))
    world = _bvh_chain(d, refs)
    desc = d.finish(world, [])
    cam = camera((4, 6, 5), (0, 0, 0), vfov=35.0, aspect=1.0, aperture=0.1, focus=8.0)
    return d, desc, cam, params(64, 64, 8, integrator=ffi.VK_INTEGRATOR_SCATTER, background=ffi.VK_BACKGROUND_SKY)


def noise_everywhere():
    """Perlin noise textures wherever one can sit: on more spheres than the lineariser's noise-sphere list holds (the device then
    looks the material up), on a rect, on a Boxy, on a sphere under a transform, behind a checker, inside a SpecDiffuse and on a
    moving sphere; two Perlin tables.  The device works the turbulence of plain sphere hits out ahead of the material code, seven
    octaves on seven lanes (vk_kernels.h cooperative_turb), and must fall back to the serial loop for all the others."""
    d = Desc()
    lights = []
    refs = _room(d, lights)
    n1, n2 = d.noise(4.0, 1), d.noise(0.7, 2)
    for k in range(6):
        refs.append(d.sphere((1.5 + 1.4 * k, 1.0 + 0.3 * k, 3.0 + 0.5 * (k % 3)), 0.7, d.mat(ffi.VK_MAT_LAMBERTIAN, n1 if k % 2 else n2)))
    refs.append(d.rect(2.0, 8.0, 2.0, 8.0, 9.5, (0, 1, 2), d.mat(ffi.VK_MAT_LAMBERTIAN, n1)))
    refs.append(d.boxy((0.5, 0.0, 6.0), (2.5, 2.0, 8.0), d.mat(ffi.VK_MAT_LAMBERTIAN, n2)))
    refs.append(d.translate(d.rotate(d.sphere((0, 0, 0), 1.0, d.mat(ffi.VK_MAT_METAL, n1, 0.2)), 1, 30.0), (7.5, 6.5, 6.0)))
    refs.append(d.sphere((5.0, 7.5, 5.0), 1.0, d.mat(ffi.VK_MAT_LAMBERTIAN, d.checker(n2, d.solid(0.9, 0.2, 0.2)))))
    sd = d.mat(ffi.VK_MAT_SPEC_DIFFUSE, 0, 0.5, d.mat(ffi.VK_MAT_METAL, d.solid(0.9, 0.9, 0.9), 0.0), d.mat(ffi.VK_MAT_LAMBERTIAN, n1))
    refs.append(d.sphere((2.5, 6.5, 4.0), 1.0, sd))
    refs.append(d.moving_sphere((8, 2, 3), (8.5, 2, 3), 0.0, 1.0, 0.8, d.mat(ffi.VK_MAT_LAMBERTIAN, n2)))
    world = _bvh_chain(d, refs)
    desc = d.finish(world, lights)
    cam = camera((5, 5, -12), (5, 5, 0), vfov=40.0)
    return d, desc, cam, params(64, 64, 8)


ALL = {"nested_transforms": nested_transforms, "noise_everywhere": noise_everywhere, "lights_and_specdiffuse": lights_and_specdiffuse,
       "media_and_textures": media_and_textures, "scatter_sky": scatter_sky, "metal_only": metal_only,
       "one_metal_among_many": one_metal_among_many}
