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


def nextweek_regions(img_bottom_up, quantise=True):
    """A 900x900 linear render of final_scene() (scene.rs:732-874, HEAD's PDF integrator) against the parts of
    sample/thenextweek.png that do not — or only weakly — depend on the reference's unseeded draws
    (tests/golden/make_nextweek_regions.py explains every region).  Returns {region: stats}:
      rel        relative error of the region's mean linear colour, per channel
      z_rms      rms over the region's BxB blocks (clamped ones excluded) of the block-mean difference in units of the combined
                 Monte-Carlo standard error of the two images (per-pixel noise measured from each image itself)
      lumvar     variance of the luminance over the region: (ours, the PNG's)
    plus light_quad (silhouette mismatch), earth_upper per-pixel code statistics and the moving sphere's blur profile.
    quantise=False compares the render's linear means as they are: Vec3::to_color clamps every PIXEL at 0.999, which at a few
    samples per pixel cuts the rare bright samples of the light-sampling estimator out of the means (the 4-spp oracle image's
    fog is 14 % darker after the clamp); the PNG, at thousands of samples per pixel, loses nothing outside its clamped blocks."""
    g = np.load(os.path.join(GOLDEN, "nextweek_regions.npz"))
    assert img_bottom_up.shape == (int(g["height"]), int(g["width"]), 3)
    B = int(g["block"])
    q = to_color(img_bottom_up)[::-1]                       # the codes the reference's PPM writer would emit (main.rs:209-211)
    lin = ((q.astype(np.float64) + 0.5) / 256.0) ** 2      # compared after the same quantisation the PNG went through
    if not quantise:
        lin = img_bottom_up[::-1].astype(np.float64)
    out = {}
    for k in g.files:
        if not k.startswith("mean_"):
            continue
        n = k[5:]
        r0, r1, c0, c1 = g["rect_" + n]
        blk = lin[r0:r1, c0:c1].reshape((r1 - r0) // B, B, (c1 - c0) // B, B, 3).transpose(0, 2, 1, 3, 4)
        mean = blk.mean((2, 3))
        sigma = neighbour_sigma(blk)
        keep = ~g["clamped_" + n]
        se = np.sqrt(sigma ** 2 + g["sigma_" + n] ** 2) / B
        z = ((mean - g[k]) / np.maximum(se, 1e-7))[keep]
        m_ours, m_ref = mean[keep].mean(0), g[k][keep].mean(0)
        # standard error of the REGION mean (all kept blocks), for renders with few samples per pixel
        se_region = np.sqrt((sigma[keep] ** 2 + g["sigma_" + n][keep] ** 2).mean(0) / (keep.sum() * B * B))
        out[n] = {"rel": (np.abs(m_ours - m_ref) / m_ref).tolist(), "z_region": (np.abs(m_ours - m_ref) / se_region).tolist(),
                  "z_rms": float(np.sqrt((z ** 2).mean())), "mean": m_ours.tolist(), "ref": m_ref.tolist(),
                  "lumvar": (float(lin[r0:r1, c0:c1].mean(2).var()), float(g["lumvar_" + n]))}
    r0, r1, c0, c1 = g["rect_light_quad"]
    ref_mask = np.unpackbits(g["mask_light_quad"])[:(r1 - r0) * (c1 - c0)].reshape(r1 - r0, c1 - c0).astype(bool)
    mask = (q[r0:r1, c0:c1] >= 255).all(2)
    out["light_quad"] = {"ref_pixels": int(ref_mask.sum()), "pixels": int(mask.sum()), "mismatch": int((mask ^ ref_mask).sum())}
    r0, r1, c0, c1 = g["rect_earth_upper"]
    a, b = q[r0:r1, c0:c1].astype(np.float64), g["px_earth_upper"].astype(np.float64)
    out["earth_upper"]["px_corr"] = [float(np.corrcoef(a[..., c].ravel(), b[..., c].ravel())[0, 1]) for c in range(3)]
    out["earth_upper"]["px_median_abs_code"] = float(np.median(np.abs(a - b)))
    r0, r1, c0, c1 = g["rect_moving_profile"]
    prof, ref_prof = lin[r0:r1, c0:c1].mean(0)[:, 0], g["profile_moving"][:, 0]     # red channel along x

    def extent(p):      # columns where the profile exceeds half its maximum: the sphere's width + the shutter's blur
        on = np.nonzero(p > 0.5 * p.max())[0]
        return int(on[0]), int(on[-1])
    out["moving_sphere"]["half_max_cols"] = (extent(prof), extent(ref_prof))
    out["moving_sphere"]["profile_corr"] = float(np.corrcoef(prof, ref_prof)[0, 1])
    return out


def check_nextweek_regions(rep, spp):
    """Tolerances: a relative floor for what the reference's unseeded scene draws move (box heights light the fog and the spheres
    from below; sphere positions; Perlin tables) plus 4 standard errors of the region mean for renders with few samples."""
    def mean_ok(n, rel_floor):
        r = rep[n]
        for c in range(3):
            assert r["rel"][c] < rel_floor or r["z_region"][c] < 4.0, (n, c, r)
    # deterministic up to indirect light off the random-height boxes
    for n in ("haze_upper_right", "wall_mid", "earth_upper"):
        mean_ok(n, 0.01)
    mean_ok("haze_left_top", 0.03)          # ~0.001 linear: a quarter of one 8-bit code
    mean_ok("moving_sphere", 0.015)
    # depend on unseeded draws through their own surface or what they mirror
    mean_ok("perlin_sphere", 0.03)
    mean_ok("metal_sphere", 0.05)
    mean_ok("sphere_cube", 0.05)
    mean_ok("blue_sphere", 0.12)            # r, g ~ 0.003-0.007 linear (1-2 codes); the blue channel carries the medium's albedo
    assert rep["blue_sphere"]["rel"][2] < 0.03 or rep["blue_sphere"]["z_region"][2] < 4.0, rep["blue_sphere"]
    # block means within Monte-Carlo noise where nothing unseeded is in view
    # (at a few samples per pixel the light-sampling estimator's rare bright samples make block means heavy-tailed: looser
    # there, and the near-black corner — a handful of such samples per block — only when converged)
    converged = spp >= 256
    for n in ("haze_upper_right", "wall_mid") + (("haze_left_top",) if converged else ()):
        assert rep[n]["z_rms"] < (1.5 if converged else 2.0), (n, rep[n])
    assert rep["moving_sphere"]["z_rms"] < 2.0, rep["moving_sphere"]
    # silhouette of the light: Camera + Rect::hit; edge pixels are Monte-Carlo (a pixel saturates when most of its samples hit)
    lq = rep["light_quad"]
    assert lq["mismatch"] < 0.01 * lq["ref_pixels"], lq
    # motion blur: the half-maximum extent of the sphere's profile along x within 3 px of the PNG's at both ends
    (a0, a1), (b0, b1) = rep["moving_sphere"]["half_max_cols"]
    assert abs(a0 - b0) <= 3 and abs(a1 - b1) <= 3, rep["moving_sphere"]
    assert rep["moving_sphere"]["profile_corr"] > (0.995 if converged else 0.97), rep["moving_sphere"]
    if converged:       # per-pixel statistics need a converged render
        assert min(rep["earth_upper"]["px_corr"]) > 0.99, rep["earth_upper"]
        assert rep["earth_upper"]["px_median_abs_code"] <= 2.0, rep["earth_upper"]
        ours, ref = rep["perlin_sphere"]["lumvar"]
        assert abs(ours - ref) / ref < 0.05, rep["perlin_sphere"]
