"""Generates tests/golden/cornell_blocks.json from the reference's only deterministic-scene
artefact, sample/therestofyourlife.png (HEAD cornell_box(), scene.rs:630-730, rendered by
HEAD main.rs: 900x900, background 0).  The PNG stores Vec3::to_color output (vec3.rs:54-61:
sqrt gamma, clamp 0.999, *256, truncate), so linear radiance ~ ((v + 0.5) / 256)^2.
Run in the build container (the reference is not present on the GPU box):
    python tests/golden/make_cornell_blocks.py
"""
import json
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/sample/therestofyourlife.png"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_blocks.json")

img = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float64)
h, w, _ = img.shape
lin = ((img + 0.5) / 256.0) ** 2
nb = 6
bh, bw = h // nb, w // nb
blocks = [[lin[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw].reshape(-1, 3).mean(0).tolist() for c in range(nb)] for r in range(nb)]
json.dump({
    "source": "reference sample/therestofyourlife.png (900x900 RGB8), rows top->bottom",
    "width": w, "height": h, "blocks": nb,
    "transform": "linear = ((png + 0.5)/256)^2  (inverse of Vec3::to_color, vec3.rs:54-61)",
    "mean_linear_rgb": lin.reshape(-1, 3).mean(0).tolist(),
    "block_mean_linear_rgb": blocks,
}, open(OUT, "w"), indent=1)
print("wrote", OUT, "mean", lin.reshape(-1, 3).mean(0))

# ---- finer fixture (round 2): 30x30 blocks of 30x30 px with the PNG's own Monte-Carlo noise, so that a test can
# ask for agreement WITHIN the noise of the two images instead of a flat percentage:
#   mean[r][c]   block mean, linear RGB (code 0 -> exactly 0: the 21-px border is background 0, main.rs:124)
#   sigma[r][c]  per-pixel noise of the block from horizontal neighbour differences, sqrt(mean(d^2)/2)
#                (signal gradients only make it larger => conservative)
#   saturated    block contains a pixel clamped by to_color (code 255): means there are biased low
B = 30
nbf = h // B
lin0 = lin.copy()
lin0[img == 0] = 0.0
blk = lin0.reshape(nbf, B, nbf, B, 3).transpose(0, 2, 1, 3, 4)
mean30 = blk.mean((2, 3))
sigma30 = np.sqrt(((blk[:, :, :, 1:] - blk[:, :, :, :-1]) ** 2).mean((2, 3)) / 2.0)
sat30 = (img.reshape(nbf, B, nbf, B, 3).transpose(0, 2, 1, 3, 4) >= 255).any((2, 3, 4))
OUT30 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_blocks30.npz")
np.savez_compressed(OUT30, block=np.int32(B), mean=mean30, sigma=sigma30, saturated=sat30)
print("wrote", OUT30, os.path.getsize(OUT30), "bytes;", int(sat30.sum()), "saturated blocks")
