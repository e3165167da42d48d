"""
Seeded synthetic Solar-Orbiter-like scenes: one analytic "truth" field in the
helioprojective plane (sum of Gaussian blobs + floor) rendered through an HRIEUV-like header (the image to
align, with a known injected pointing error) and through an FSI-like header (the reference image).
No FITS files, no network.  numpy only.
"""
from __future__ import annotations

import numpy as np

AU = 1.495978707e11


def _header(naxis1, naxis2, crpix1, crpix2, crval1, crval2, cdelt1, cdelt2, crota, unit="arcsec", dsun_au=0.38,
            crln=250.0, crlt=-3.0, wavelnth=174, date="2022-03-17T09:50:45.277"):
    rho = np.deg2rad(crota)
    lam = cdelt2 / cdelt1
    return {
        "NAXIS": 2, "NAXIS1": int(naxis1), "NAXIS2": int(naxis2),
        "CTYPE1": "HPLN-TAN", "CTYPE2": "HPLT-TAN", "CUNIT1": unit, "CUNIT2": unit,
        "CRPIX1": float(crpix1), "CRPIX2": float(crpix2), "CRVAL1": float(crval1), "CRVAL2": float(crval2),
        "CDELT1": float(cdelt1), "CDELT2": float(cdelt2),
        "PC1_1": float(np.cos(rho)), "PC1_2": float(-lam * np.sin(rho)),
        "PC2_1": float(np.sin(rho) / lam), "PC2_2": float(np.cos(rho)),
        "CROTA": float(crota), "LONPOLE": 180.0,
        "DSUN_OBS": float(dsun_au * AU), "CRLN_OBS": float(crln), "CRLT_OBS": float(crlt),
        "WAVELNTH": int(wavelnth), "DATE-AVG": date, "DATE-OBS": date,
    }


def _pixel_to_plane(hdr):
    """Pixel -> helioprojective plane coordinates (arcsec), small-angle linear form of the header."""
    u = {"arcsec": 1.0, "deg": 3600.0}[hdr["CUNIT1"]]
    y, x = np.mgrid[0:hdr["NAXIS2"], 0:hdr["NAXIS1"]].astype(np.float64)
    q1 = x + 1.0 - hdr["CRPIX1"]
    q2 = y + 1.0 - hdr["CRPIX2"]
    tx = hdr["CRVAL1"] * u + hdr["CDELT1"] * u * (hdr["PC1_1"] * q1 + hdr["PC1_2"] * q2)
    ty = hdr["CRVAL2"] * u + hdr["CDELT2"] * u * (hdr["PC2_1"] * q1 + hdr["PC2_2"] * q2)
    return tx, ty


def _plane_to_pixel(hdr, tx, ty):
    u = {"arcsec": 1.0, "deg": 3600.0}[hdr["CUNIT1"]]
    m = np.array([[hdr["CDELT1"] * u * hdr["PC1_1"], hdr["CDELT1"] * u * hdr["PC1_2"]],
                  [hdr["CDELT2"] * u * hdr["PC2_1"], hdr["CDELT2"] * u * hdr["PC2_2"]]])
    mi = np.linalg.inv(m)
    dx, dy = tx - hdr["CRVAL1"] * u, ty - hdr["CRVAL2"] * u
    return mi[0, 0] * dx + mi[0, 1] * dy + hdr["CRPIX1"] - 1.0, mi[1, 0] * dx + mi[1, 1] * dy + hdr["CRPIX2"] - 1.0


def _render(hdr, blobs, floor, rng, noise=True):
    """Sum of Gaussian blobs (centre tx, ty [arcsec], sigma [arcsec], amplitude) on the pixel grid of hdr."""
    ny, nx = hdr["NAXIS2"], hdr["NAXIS1"]
    img = np.full((ny, nx), float(floor))
    u = {"arcsec": 1.0, "deg": 3600.0}[hdr["CUNIT1"]]
    scale = min(abs(hdr["CDELT1"]), abs(hdr["CDELT2"])) * u  # arcsec / px
    tx, ty = _pixel_to_plane(hdr)
    cx, cy = _plane_to_pixel(hdr, blobs[:, 0], blobs[:, 1])
    for k in range(blobs.shape[0]):
        r = int(np.ceil(4.5 * blobs[k, 2] / scale)) + 1
        x0, x1 = int(np.floor(cx[k])) - r, int(np.floor(cx[k])) + r + 1
        y0, y1 = int(np.floor(cy[k])) - r, int(np.floor(cy[k])) + r + 1
        x0, x1, y0, y1 = max(x0, 0), min(x1, nx), max(y0, 0), min(y1, ny)
        if x0 >= x1 or y0 >= y1:
            continue
        d2 = (tx[y0:y1, x0:x1] - blobs[k, 0]) ** 2 + (ty[y0:y1, x0:x1] - blobs[k, 1]) ** 2
        img[y0:y1, x0:x1] += blobs[k, 3] * np.exp(-d2 / (2.0 * blobs[k, 2] ** 2))
    if noise:
        img = img + np.sqrt(img) * rng.standard_normal(img.shape)
    return img


def make_scene(small_n=2048, large_n=3072, seed=20220317, n_blobs=400, pointing_error=(17.0, -9.0, 0.3),
               nan_frac=0.005, float32_exact=True, small_shape=None, small_cdelt=None, small_unit="arcsec",
               large_cdelt=None, large_crval=(12.5, -7.25)):
    """Returns (data_small, hdr_small, data_large, hdr_large, truth).

    The small image is rendered through the TRUE header (CRVAL = (-310, 420) arcsec, CROTA 3.3 deg for the
    default error) and handed out with a WRONG header (CRVAL - (17, -9) arcsec, CROTA 3.0): the sweep should
    recover lag ~ (+17, -9) arcsec, +0.3 deg.  `float32_exact` rounds pixel values to float32 first, as L2 FITS
    data (BITPIX=-32) cast to float64 by the reference (alignment.py:301,314) are."""
    rng = np.random.default_rng(seed)
    fov = 2048 * 0.492  # arcsec
    if small_shape is None:
        small_shape = (small_n, small_n)
    sny, snx = small_shape
    if small_cdelt is None:
        small_cdelt = (fov / snx, fov / sny)
    lcd = 3072 * 4.44 / large_n if large_cdelt is None else float(large_cdelt)
    true_crval = (-310.0, 420.0)
    true_crota = 3.0 + pointing_error[2]
    u = {"arcsec": 1.0, "deg": 1.0 / 3600.0}[small_unit]
    hdr_true = _header(snx, sny, (snx + 1) / 2.0, (sny + 1) / 2.0, true_crval[0] * u, true_crval[1] * u,
                       small_cdelt[0] * u, small_cdelt[1] * u, true_crota, unit=small_unit)
    hdr_small = _header(snx, sny, (snx + 1) / 2.0, (sny + 1) / 2.0, (true_crval[0] - pointing_error[0]) * u,
                        (true_crval[1] - pointing_error[1]) * u, small_cdelt[0] * u, small_cdelt[1] * u, 3.0,
                        unit=small_unit)
    hdr_large = _header(large_n, large_n, (large_n + 1) / 2.0, (large_n + 1) / 2.0, large_crval[0], large_crval[1], lcd, lcd, 0.0,
                        wavelnth=174, date="2022-03-17T09:50:45.281")
    half = 0.5 * max(small_cdelt[0] * snx, small_cdelt[1] * sny) + 150.0
    blobs = np.empty((n_blobs, 4))
    blobs[:, 0] = true_crval[0] + rng.uniform(-half, half, n_blobs)
    blobs[:, 1] = true_crval[1] + rng.uniform(-half, half, n_blobs)
    blobs[:, 2] = rng.uniform(3.0, 40.0, n_blobs) * 0.492  # sigma 3..40 HRI pixels, in arcsec
    blobs[:, 3] = np.exp(rng.uniform(np.log(50.0), np.log(3000.0), n_blobs))
    small = _render(hdr_true, blobs, 100.0, rng)
    large = _render(hdr_large, blobs, 100.0, rng)
    if float32_exact:
        small = small.astype(np.float32).astype(np.float64)
        large = large.astype(np.float32).astype(np.float64)
    if nan_frac > 0:
        small[rng.random(small.shape) < nan_frac] = np.nan
    truth = {"lag_crval1": pointing_error[0], "lag_crval2": pointing_error[1], "lag_crota": pointing_error[2],
             "blobs": blobs}
    return small, hdr_small, large, hdr_large, truth


def make_series(n_frames=5, n=256, seed=7, n_blobs=200, jitter_sigma=1.5, cadence_s=5.0, nan_frac=0.002,
                jitters=None):
    """Jittering time series of one HRIEUV-like field (the input of jitter_correction_imagers): every frame is rendered
    through its TRUE header (nominal CRVAL + jitter_k) and handed out, as float32 pixels, with the NOMINAL header.
    Frame 0 has no jitter.  Returns ([(data_f32, header), ...], jitters[n_frames, 2] in arcsec)."""
    rng = np.random.default_rng(seed)
    fov = 2048 * 0.492
    cd = fov / n
    crval = (-310.0, 420.0)
    if jitters is None:
        jitters = np.round(rng.normal(0.0, jitter_sigma, size=(n_frames, 2)), 2)
        jitters[0] = 0.0
    jitters = np.asarray(jitters, dtype=np.float64)
    half = 0.5 * fov + 100.0
    blobs = np.empty((n_blobs, 4))
    blobs[:, 0] = crval[0] + rng.uniform(-half, half, n_blobs)
    blobs[:, 1] = crval[1] + rng.uniform(-half, half, n_blobs)
    blobs[:, 2] = rng.uniform(3.0, 40.0, n_blobs) * 0.492
    blobs[:, 3] = np.exp(rng.uniform(np.log(50.0), np.log(3000.0), n_blobs))
    frames = []
    for k in range(n_frames):
        sec = 45.277 + cadence_s * k
        date = "2022-03-17T09:%02d:%06.3f" % (50 + int(sec // 60), sec % 60.0)
        true = _header(n, n, (n + 1) / 2.0, (n + 1) / 2.0, crval[0] + jitters[k, 0], crval[1] + jitters[k, 1], cd, cd,
                       3.0, date=date)
        nominal = _header(n, n, (n + 1) / 2.0, (n + 1) / 2.0, crval[0], crval[1], cd, cd, 3.0, date=date)
        img = _render(true, blobs, 100.0, rng).astype(np.float32)
        if nan_frac > 0:
            img[rng.random(img.shape) < nan_frac] = np.nan
        frames.append((img, nominal))
    return frames, jitters


def make_spice_l2(nx=48, ny=160, nw=16, large_n=192, seed=21, pointing_error=(-23.0, 36.0, 0.0), n_blobs=150):
    """SPICE-L2-like raster window: 4-D cube [1, nw, ny, nx] (float32) whose sum over wavelength is a helioprojective
    image with CDELT (4.0, 1.098) arcsec and a known pointing error, plus the FSI-like reference image.
    Returns (cube, hdr4d, data_large, hdr_large, truth); truth also holds the 2-D image and its (wrong) header."""
    small, hs, large, hl, truth = make_scene(small_n=ny, large_n=large_n, seed=seed, n_blobs=n_blobs,
                                             pointing_error=pointing_error, nan_frac=0.0, small_shape=(ny, nx),
                                             small_cdelt=(4.0, 1.098), float32_exact=False, large_cdelt=4.44,
                                             large_crval=(-300.0, 400.0))
    k = np.arange(nw) - (nw - 1) / 2.0
    prof = np.exp(-0.5 * (k / 2.0) ** 2)
    prof /= prof.sum()
    cube = (small[None, :, :] * prof[:, None, None])[None].astype(np.float32)
    h = dict(hs)
    h.update({"NAXIS": 4, "NAXIS1": nx, "NAXIS2": ny, "NAXIS3": nw, "NAXIS4": 1, "CTYPE3": "WAVE", "CTYPE4": "TIME",
              "CUNIT3": "nm", "CUNIT4": "s", "CRPIX3": (nw + 1) / 2.0, "CRPIX4": 1.0, "CRVAL3": 97.7031,
              "CRVAL4": 600.0, "CDELT3": 0.00973, "CDELT4": 1.0, "PC3_3": 1.0, "PC4_4": 1.0, "PC4_1": -25.2,
              "NBIN2": 4, "DETECTOR": "SW", "PXBEG2": 200, "SOLAR_B0": -3.0, "RSUN_REF": 695700000.0,
              "DATE-BEG": "2022-03-17T09:40:45.277", "DATEREF": "2022-03-17T09:40:45.277", "TIMESYS": "UTC"})
    truth = dict(truth, image=small, header2d=hs, profile=prof)
    return cube, h, large, hl, truth


def make_imager_sequence(large, hdr_large, start="2022-03-17T09:40:00.000", cadence_s=300.0, n_frames=6):
    """Imager frames for the synthetic-raster builder: the same FSI-like image with a frame-dependent gain
    (1 + 0.05 k), pointing drift (0.7 k, -0.4 k) arcsec and DATE-AVG = start + k * cadence.  float32 pixels."""
    import datetime as dt
    t0 = dt.datetime.strptime(start, "%Y-%m-%dT%H:%M:%S.%f")
    frames = []
    for k in range(n_frames):
        h = dict(hdr_large)
        h["CRVAL1"] = hdr_large["CRVAL1"] + 0.7 * k
        h["CRVAL2"] = hdr_large["CRVAL2"] - 0.4 * k
        t = t0 + dt.timedelta(seconds=cadence_s * k)
        h["DATE-AVG"] = t.strftime("%Y-%m-%dT%H:%M:%S.%f")[:-3]
        h["DETECTOR"] = "FSI"
        frames.append(((large * (1.0 + 0.05 * k)).astype(np.float32), h))
    return frames


def make_car_scene(small_shape=(90, 120), large_shape=(150, 200), seed=31, n_blobs=160, pointing_error=(0.018, -0.011),
                   small_cdelt=(0.0101, 0.0099), large_cdelt=(0.01637, 0.01613), crota=0.0, explicit_lonpole=False,
                   nan_frac=0.004):
    """Two Carrington maps (CRLN-CAR / CRLT-CAR, plate carree, degrees) of one blob field on the sphere: the map to
    align is rendered through its TRUE header and handed out with CRVAL off by `pointing_error` (degrees), the reference
    map is coarser and wider.  float32 pixels.  Returns (small, hdr_small, large, hdr_large, truth)."""
    rng = np.random.default_rng(seed)

    def header(shape, crval, cdelt, rot):
        ny, nx = shape
        rho, lam = np.deg2rad(rot), cdelt[1] / cdelt[0]
        h = {"NAXIS": 2, "NAXIS1": int(nx), "NAXIS2": int(ny), "CTYPE1": "CRLN-CAR", "CTYPE2": "CRLT-CAR",
             "CUNIT1": "deg", "CUNIT2": "deg", "CRPIX1": (nx + 1) / 2.0, "CRPIX2": (ny + 1) / 2.0,
             "CRVAL1": float(crval[0]), "CRVAL2": float(crval[1]), "CDELT1": float(cdelt[0]), "CDELT2": float(cdelt[1]),
             "PC1_1": float(np.cos(rho)), "PC1_2": float(-lam * np.sin(rho)), "PC2_1": float(np.sin(rho) / lam),
             "PC2_2": float(np.cos(rho)), "CROTA": float(rot), "DATE-AVG": "2022-03-17T09:50:45.277", "WAVELNTH": 174}
        if explicit_lonpole:
            h["LONPOLE"] = 0.0
            h["LATPOLE"] = 90.0
        return h

    centre = (250.0, 0.0)
    half = 0.5 * max(large_shape[1] * large_cdelt[0], large_shape[0] * large_cdelt[1])
    blobs = np.empty((n_blobs, 4))
    blobs[:, 0] = centre[0] + rng.uniform(-half, half, n_blobs)
    blobs[:, 1] = centre[1] + rng.uniform(-half, half, n_blobs)
    blobs[:, 2] = rng.uniform(0.02, 0.25, n_blobs)  # sigma, degrees
    blobs[:, 3] = np.exp(rng.uniform(np.log(50.0), np.log(3000.0), n_blobs))

    def render(h):
        # pixel -> (lon, lat): with |CRVAL2| << 1 deg the oblique terms are far below a pixel; the affine form is used
        # for rendering only (the alignment itself uses the exact projection)
        y, x = np.mgrid[0:h["NAXIS2"], 0:h["NAXIS1"]].astype(np.float64)
        q1, q2 = x + 1.0 - h["CRPIX1"], y + 1.0 - h["CRPIX2"]
        lon = h["CRVAL1"] + h["CDELT1"] * (h["PC1_1"] * q1 + h["PC1_2"] * q2)
        lat = h["CRVAL2"] + h["CDELT2"] * (h["PC2_1"] * q1 + h["PC2_2"] * q2)
        img = np.full(lon.shape, 100.0)
        for k in range(n_blobs):
            d2 = (lon - blobs[k, 0]) ** 2 + (lat - blobs[k, 1]) ** 2
            m = d2 < (4.5 * blobs[k, 2]) ** 2
            img[m] += blobs[k, 3] * np.exp(-d2[m] / (2.0 * blobs[k, 2] ** 2))
        img = img + np.sqrt(img) * rng.standard_normal(img.shape)
        return img.astype(np.float32)

    true_crval = (centre[0] + 0.0713, centre[1] + 0.00037)
    h_true = header(small_shape, true_crval, small_cdelt, crota)
    h_small = header(small_shape, (true_crval[0] - pointing_error[0], true_crval[1] - pointing_error[1]), small_cdelt,
                     crota)
    h_large = header(large_shape, centre, large_cdelt, 0.0)
    small, large = render(h_true), render(h_large)
    if nan_frac > 0:
        small[rng.random(small.shape) < nan_frac] = np.nan
    truth = {"lag_crval1": pointing_error[0], "lag_crval2": pointing_error[1]}
    return small, h_small, large, h_large, truth
