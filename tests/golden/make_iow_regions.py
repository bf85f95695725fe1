"""Generates tests/golden/iow_regions.npz from the reference's sample/inoneweekend.png (1024x576 RGB8),
the only artefact of the InOneWeekend state of the reference (README.md:5-7; the tag itself is not
readable).  The random small spheres of that render are unseeded, but parts of the picture do not
depend on them and pin the scatter integrator, the sky, the IOW camera (incl. its defocus blur at the
horizon), Metal with fuzz 0, Lambertian::scatter's cosine law and Vec3::to_color:

  per-pixel regions (every pixel a deterministic function of the scene, up to +-1 code of sampling noise):
    sky_left, sky_right   sky gradient (1-t)*1 + t*(0.5,0.7,1.0) seen through the camera, to_color'd (vec3.rs:54-61)
    metal_cap             upper half of the r=1 Metal(0.7,0.6,0.5, fuzz 0) sphere at (4,1,0) (scene.rs:236-240):
                          reflect() of the sky times the albedo
  mean regions (Monte-Carlo; compared through block means with the noise measured from the images):
    ground_far            the r=1000 Lambertian(0.5) ground beyond the small spheres (scene.rs:176-182): convex, so its
                          radiance is 0.5 * the cosine-weighted sky = (0.2917, 0.375, 0.5) analytically
    brown_sphere          visible part of the Lambertian(0.4,0.2,0.1) r=1 sphere at (-4,1,0) (scene.rs:232-235)
    glass_lower           lower half of the Dielectric(1.5) r=1 sphere at (0,1,0) (scene.rs:226-228): refracted ground/sky

Region rectangles are (row0, row1, col0, col1) in PNG coordinates (row 0 = top), chosen by eye on the PNG.
Run in the build container (the reference is not present on the GPU box):
    python tests/golden/make_iow_regions.py
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/sample/inoneweekend.png"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "iow_regions.npz")

PER_PIXEL = {"sky_left": (0, 120, 0, 320), "sky_right": (0, 120, 860, 1024), "metal_cap": (70, 190, 560, 780)}
MEAN = {"ground_far": (137, 150, 0, 300), "brown_sphere": (70, 190, 345, 395), "glass_lower": (175, 235, 410, 480)}

img = np.asarray(Image.open(SRC).convert("RGB"))
assert img.shape == (576, 1024, 3)
out = {"width": np.int32(1024), "height": np.int32(576)}
for name, (r0, r1, c0, c1) in PER_PIXEL.items():
    out["px_" + name] = img[r0:r1, c0:c1].copy()
    out["rect_" + name] = np.array([r0, r1, c0, c1], np.int32)
lin = ((img.astype(np.float64) + 0.5) / 256.0) ** 2        # inverse of Vec3::to_color (vec3.rs:54-61)
for name, (r0, r1, c0, c1) in MEAN.items():
    blk = lin[r0:r1, c0:c1]
    # per-pixel Monte-Carlo noise from horizontal neighbour differences (signal gradients only make it larger)
    sigma = np.sqrt(((blk[:, 1:] - blk[:, :-1]) ** 2).mean((0, 1)) / 2.0)
    out["mean_" + name] = blk.reshape(-1, 3).mean(0)
    out["sigma_" + name] = sigma
    out["rect_" + name] = np.array([r0, r1, c0, c1], np.int32)
    print(name, out["mean_" + name], "sigma/pixel", sigma, "n", blk.shape[0] * blk.shape[1])
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
