"""Decodes the reference's texture images (assets/*.png, read by ImageTexture::new at material.rs:269-279 from
scene.rs:230,355-398,588,822) to gzip'd binary PPM (P6: the decoded 8-bit RGB bytes, row 0 = top, exactly the buffer
`png::Reader::next_frame` fills) under tests/golden/assets/.  These are DATA fixtures: the PNG decode (`png 0.16.6`) is
third-party and lossless, so any correct decoder yields the same bytes; the host mirror (vecchio_amd/host) reads the .ppm.gz
where the reference reads the .png.  Run in the build container (the reference is not present on the GPU box):
    python tests/golden/make_assets.py
"""
import gzip
import hashlib
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/assets"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
os.makedirs(OUT, exist_ok=True)
for name in ("earthmap", "bowser_face", "bowser_top", "bowser_back", "bowser_side", "twitter"):
    im = Image.open(os.path.join(SRC, name + ".png"))
    assert im.mode == "RGB", (name, im.mode)          # 8-bit RGB, BPP = 3 (material.rs:267)
    a = np.asarray(im)
    h, w, _ = a.shape
    raw = b"P6\n%d %d\n255\n" % (w, h) + a.tobytes()
    path = os.path.join(OUT, name + ".ppm.gz")
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", compresslevel=9, mtime=0, filename="") as g:
            g.write(raw)
    print(f"{name}: {w}x{h}, {len(raw)} bytes raw, {os.path.getsize(path)} gz, sha256(rgb) {hashlib.sha256(a.tobytes()).hexdigest()[:16]}")
