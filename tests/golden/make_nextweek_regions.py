"""Generates tests/golden/nextweek_regions.npz from the reference's sample/thenextweek.png (900x900 RGB8; README.md:9-15),
the only reference-held artefact that shows ConstantMedium + Isotropic (hittable.rs:436-498, material.rs:436-465), MovingSphere
(hittable.rs:136-197), Perlin / NoiseTexture (material.rs:306-434), ImageTexture + Sphere uv (material.rs:261-304,
hittable.rs:54-61) and the fuzz-10 Metal (scene.rs:793-796) — i.e. what distinguishes C3 from C4.

Which code rendered it?  The picture's objects are HEAD's final_scene() (scene.rs:732-874) at HEAD's camera; rendered both ways
by the HIP path (900x900, 2000 spp), HEAD's PDF integrator (main.rs:123-153) reproduces the PNG's fog haze, walls and spheres to
0.1-0.5 % while a plain scatter integrator (emitted + attenuation * L) is 20-60 % off in the haze: the PNG on master was made by
the PDF integrator.  So the scene compared is `final_scene` as it stands at HEAD.

The reference is unseeded: box heights (scene.rs:751), the 1000 small spheres (scene.rs:838-843), the Perlin tables
(material.rs:357-377) and every sample differ between any two runs.  Regions (PNG coordinates: row 0 = top) are therefore
compared STATISTICALLY, through quantities that do not depend on those draws, or only weakly:

  light_quad       rows 0..140: the pixels the emitter saturates (XZRect 123..423 x 147..412 @ y 554, emit 7, scene.rs:762-769):
                   a per-pixel silhouette -> Camera::new/get_ray (main.rs:71-120) + Rect::hit (hittable.rs:229-256)
  haze_upper_right nothing but the r = 5000 fog (density 1e-4, white, scene.rs:810-819) in front of the black background
                   -> ConstantMedium::hit + Isotropic through the integrator, lit by the quad
  haze_left_top    the same, in the dark corner beside the light
  wall_mid         fog + far ground between the spheres
  moving_sphere    the motion-blurred Lambertian (0.7,0.3,0.1) sphere, centre 400..430 over the shutter (scene.rs:771-783)
                   -> MovingSphere::hit / center(), shutter time draw; its extent along x is the blur
  earth_upper      upper half of the ImageTexture sphere (scene.rs:821-826), lit directly: block means AND per-pixel codes
                   -> spherical uv (atan2/asin), v -> 1-v, texel fetch of the real earthmap
  blue_sphere      the glass sphere filled with the blue medium (density 0.2, scene.rs:799-808): centre of its disc
  perlin_sphere    NoiseTexture(0.1) sphere (scene.rs:827-832): mean and VARIANCE of the luminance (the tables are unseeded, the
                   speckle statistics are not)
  metal_sphere     Metal(0.8,0.8,0.9, fuzz 10) (scene.rs:793-796): mean colour
  sphere_cube      the 1000 r = 10 spheres, rotated 15 deg, translated (scene.rs:836-851): mean colour (positions are unseeded)

For every region the fixture holds BxB-pixel block means of the linear colour (inverse of Vec3::to_color, vec3.rs:54-61), the
PNG's per-pixel noise (neighbour differences) and which blocks contain clamped pixels.
Run in the build container (the reference is not present on the GPU box):   python tests/golden/make_nextweek_regions.py
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/sample/thenextweek.png"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nextweek_regions.npz")
B = 10
REGIONS = {   # (row0, row1, col0, col1), multiples of B
    "haze_upper_right": (20, 250, 620, 900), "haze_left_top": (0, 150, 0, 100), "wall_mid": (150, 340, 260, 440),
    "moving_sphere": (180, 330, 50, 240), "earth_upper": (430, 540, 30, 250), "blue_sphere": (600, 720, 170, 310),
    "perlin_sphere": (370, 530, 320, 480), "metal_sphere": (590, 690, 710, 820), "sphere_cube": (270, 460, 470, 700),
}
LIGHT = (0, 140, 90, 620)
MOVING_PROFILE = (230, 280, 20, 280)    # rows through the moving sphere's middle: mean over rows -> the blur profile along x

img = np.asarray(Image.open(SRC).convert("RGB"))
assert img.shape == (900, 900, 3)
lin = ((img.astype(np.float64) + 0.5) / 256.0) ** 2        # inverse of Vec3::to_color (vec3.rs:54-61)
out = {"width": np.int32(900), "height": np.int32(900), "block": np.int32(B)}
for name, (r0, r1, c0, c1) in REGIONS.items():
    assert (r1 - r0) % B == 0 and (c1 - c0) % B == 0
    blk = lin[r0:r1, c0:c1].reshape((r1 - r0) // B, B, (c1 - c0) // B, B, 3).transpose(0, 2, 1, 3, 4)
    raw = img[r0:r1, c0:c1].reshape((r1 - r0) // B, B, (c1 - c0) // B, B, 3).transpose(0, 2, 1, 3, 4)
    d = blk[:, :, :, 1:, :] - blk[:, :, :, :-1, :]
    out["rect_" + name] = np.array([r0, r1, c0, c1], np.int32)
    out["mean_" + name] = blk.mean((2, 3))
    out["sigma_" + name] = np.sqrt((d ** 2).mean((2, 3)) / 2.0)     # per-pixel noise (signal gradients only make it larger)
    out["clamped_" + name] = (raw >= 255).any((2, 3, 4))
    lum = lin[r0:r1, c0:c1].mean(2)
    out["lumvar_" + name] = np.float64(lum.var())
    print(f"{name:17s} mean {lin[r0:r1, c0:c1].reshape(-1, 3).mean(0).round(4)} sigma/px {out['sigma_' + name].mean((0, 1)).round(4)} "
          f"clamped blocks {int(out['clamped_' + name].sum())} lum var {lum.var():.5f}")
r0, r1, c0, c1 = LIGHT
out["rect_light_quad"] = np.array(LIGHT, np.int32)
out["mask_light_quad"] = np.packbits((img[r0:r1, c0:c1] >= 255).all(2))
print("light_quad saturated pixels", int((img[r0:r1, c0:c1] >= 255).all(2).sum()))
r0, r1, c0, c1 = REGIONS["earth_upper"]
out["px_earth_upper"] = img[r0:r1, c0:c1].copy()
r0, r1, c0, c1 = MOVING_PROFILE
out["rect_moving_profile"] = np.array(MOVING_PROFILE, np.int32)
out["profile_moving"] = lin[r0:r1, c0:c1].mean(0)              # (cols, 3)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
