"""Comparisons of a rendered image with the fixtures made from the reference's own sample PNGs
(tests/golden/make_*.py).  Used with the oracle's image on the CPU and with the HIP path's image on the
GPU, so both are pinned to the same reference artefacts.  All of these are STATISTICAL: the reference is
unseeded (rand::thread_rng), so only quantities that do not depend on its random stream are compared."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def to_color(img):
    """Vec3::to_color (vec3.rs:54-61): sqrt gamma, clamp [0, 0.999], *256, truncate (NaN -> 0 via `as u32`)."""
    v = np.sqrt(img.astype(np.float32))
    c = np.where(v < 0, np.float32(0), np.where(v > np.float32(0.999), np.float32(0.999), v))
    with np.errstate(invalid="ignore"):
        q = np.floor(np.float32(256.0) * c)
    return np.nan_to_num(q, nan=0.0).astype(np.uint8)


def neighbour_sigma(blk):
    """per-pixel noise of (..., rows, cols, 3) blocks from horizontal neighbour differences"""
    d = blk[..., :, 1:, :] - blk[..., :, :-1, :]
    return np.sqrt((d ** 2).mean((-3, -2)) / 2.0)


def cornell_blocks30_z(img_bottom_up):
    """z-scores of the 30x30-px block means of a 900x900 linear render of cornell_box() against
    sample/therestofyourlife.png (scene.rs:630-730, main.rs:28-29,171), in units of the combined Monte-Carlo
    standard error of the two block means (each image's per-pixel noise is measured from the image itself).
    Blocks the PNG clamps (the light) are excluded.  Returns (z[n,3], relative error[n,3])."""
    g = np.load(os.path.join(GOLDEN, "cornell_blocks30.npz"))
    B = int(g["block"])
    lin = img_bottom_up[::-1].astype(np.float64)          # PNG rows are top-down (main.rs:209)
    assert lin.shape == (900, 900, 3)
    nb = 900 // B
    blk = lin.reshape(nb, B, nb, B, 3).transpose(0, 2, 1, 3, 4)
    mean = blk.mean((2, 3))
    sigma = neighbour_sigma(blk)
    se = np.sqrt(sigma ** 2 + g["sigma"] ** 2) / B
    keep = ~g["saturated"]
    z = (mean - g["mean"]) / np.maximum(se, 1e-6)
    rel = np.abs(mean - g["mean"]) / (g["mean"] + 0.01)
    return z[keep], rel[keep]


def iow_regions(img_bottom_up):
    """A 1024x576 render of the InOneWeekend scene (scatter integrator, sky, IOW camera) against the
    scene-independent parts of sample/inoneweekend.png.  Returns {region: stats}:
      per-pixel regions: fraction of pixels whose 8-bit code differs by more than 1, and the largest difference
      mean regions: relative error of the mean linear colour, per channel"""
    g = np.load(os.path.join(GOLDEN, "iow_regions.npz"))
    assert img_bottom_up.shape == (int(g["height"]), int(g["width"]), 3)
    q = to_color(img_bottom_up)[::-1].astype(np.int32)
    lin = ((q + 0.5) / 256.0) ** 2
    out = {}
    for k in g.files:
        if k.startswith("px_"):
            r0, r1, c0, c1 = g["rect_" + k[3:]]
            d = np.abs(q[r0:r1, c0:c1] - g[k].astype(np.int32)).max(2)
            out[k[3:]] = {"frac_gt1": float((d > 1).mean()), "max": int(d.max())}
        elif k.startswith("mean_"):
            r0, r1, c0, c1 = g["rect_" + k[5:]]
            m = lin[r0:r1, c0:c1].reshape(-1, 3).mean(0)
            out[k[5:]] = {"rel": (np.abs(m - g[k]) / g[k]).tolist(), "mean": m.tolist(), "ref": g[k].tolist()}
    return out


def check_iow_regions(rep):
    # sky: every pixel a deterministic function of the camera and the gradient -> +-1 code everywhere
    for n in ("sky_left", "sky_right"):
        assert rep[n]["max"] <= 1, (n, rep[n])
    # metal cap: reflect(sky) * albedo; its rim reflects the (unseeded) small spheres -> 99 % of the pixels
    assert rep["metal_cap"]["frac_gt1"] < 0.01, rep["metal_cap"]
    # Monte-Carlo means; the random small spheres around them move these by a fraction of a percent
    assert max(rep["ground_far"]["rel"]) < 0.005, rep["ground_far"]
    assert max(rep["brown_sphere"]["rel"]) < 0.02, rep["brown_sphere"]
    assert max(rep["glass_lower"]["rel"]) < 0.02, rep["glass_lower"]
