#!/opt/conda/bin/python3.9
"""
Golden-vector generator: runs the REFERENCE's own `euispice_coreg/utils/rectify.py`
(CarringtonTransform + Rectifier + interpol2d) in the build container and
stores inputs + outputs as a small .npz.  The reference cannot travel to the GPU box; the .npz can.

Run (build container only; /root/reference must exist):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_rectify.py

Interpreter notes: conda python 3.9 / numpy 1.26.4 / scipy 1.7.1 / astropy 4.3.1.  numpy 1.26 is put in
NEP-50 ("weak") promotion state so that the dtype flow equals the reference's pinned numpy 2.2.6
(SURVEY 8c); two names numpy removed are shimmed so astropy 4.3.1 imports.
Only the reference module is executed; nothing of it is copied.
"""
import importlib.util
import os
import sys

import numpy as np

for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)
np._set_promotion_state("weak")

REF = "/root/reference/euispice_coreg/utils/rectify.py"
spec = importlib.util.spec_from_file_location("ref_rectify", REF)
rectify = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rectify)

AU = 1.495978707e11


def make_image(rng, ny, nx, nan_frac):
    yy, xx = np.mgrid[0:ny, 0:nx]
    img = 100.0 + 50.0 * np.sin(xx / 5.3) * np.cos(yy / 7.1) + 30.0 * rng.standard_normal((ny, nx))
    img += 400.0 * np.exp(-((xx - nx * 0.4) ** 2 + (yy - ny * 0.6) ** 2) / (2 * 6.0 ** 2))
    m = rng.random((ny, nx)) < nan_frac
    img[m] = np.nan
    return img


def run_case(name, hdr, solar_r, shape, lonlims, latlims, order, img, out):
    t = rectify.CarringtonTransform(hdr, radius_correction=solar_r, reference_date=hdr["DATE-OBS"], rate_wave=None)
    r = rectify.Rectifier(t)
    res = r(img, shape, lonlims, latlims, order=order, fill=-32762)
    x, y = r.coordinates
    nx, ny = t(x=x, y=y)
    assert nx.dtype == np.float64, nx.dtype
    res = np.where(res == -32762, np.nan, res)
    keys = ["CROTA", "CROTA2", "CRVAL1", "CRVAL2", "CRPIX1", "CRPIX2", "CDELT1", "CDELT2", "DSUN_OBS", "CRLN_OBS",
            "CRLT_OBS"]
    out[name + "/hdr_keys"] = np.array([k for k in keys if k in hdr])
    out[name + "/hdr_vals"] = np.array([float(hdr[k]) for k in keys if k in hdr])
    out[name + "/solar_r"] = np.float64(solar_r)
    out[name + "/shape"] = np.array(shape)
    out[name + "/lonlims"] = np.array(lonlims, dtype=np.float64)
    out[name + "/latlims"] = np.array(latlims, dtype=np.float64)
    out[name + "/order"] = np.int64(order)
    out[name + "/image"] = img
    out[name + "/nx"] = nx
    out[name + "/ny"] = ny
    out[name + "/resampled"] = res
    print(name, "grid", res.shape, "finite", np.isfinite(res).sum(), "of", res.size)


def main():
    rng = np.random.default_rng(20220317)
    out = {}
    base = {"CROTA": 3.0, "CRVAL1": -310.0, "CRVAL2": 420.0, "CRPIX1": 32.5, "CRPIX2": 32.5, "CDELT1": 15.7,
            "CDELT2": 15.7, "DSUN_OBS": 0.38 * AU, "CRLN_OBS": 250.0, "CRLT_OBS": -3.0,
            "DATE-OBS": "2022-03-17T09:50:45.277"}
    img = make_image(rng, 64, 64, 0.01)
    # A: HRIEUV-like, grid partly outside the FOV
    run_case("A", dict(base), 1.004, [48, 40], [228.0, 262.0], [-12.0, 22.0], 2, img, out)
    # B: rotated, anisotropic CDELT, wide grid reaching beyond the limb (zz < 0 -> NaN)
    hb = dict(base)
    hb.update({"CROTA": 31.7, "CDELT1": 40.0, "CDELT2": 27.0, "CRVAL1": 55.5, "CRVAL2": -120.25, "CRPIX1": 30.0,
               "CRPIX2": 35.0, "DSUN_OBS": 0.61 * AU})
    run_case("B", hb, 1.0, [56, 24], [120.0, 380.0], [-85.0, 85.0], 2, make_image(rng, 64, 64, 0.0), out)
    # C: order 1
    run_case("C", dict(base), 1.004, [33, 47], [236.0, 256.0], [-4.0, 16.0], 1, img, out)
    # D: CROTA2 keyword instead of CROTA, non-square image, shifted header (a "lag")
    hd = dict(base)
    del hd["CROTA"]
    hd["CROTA2"] = -12.25
    hd["CRVAL1"] += 17.0
    hd["CRVAL2"] -= 9.0
    hd["CRPIX1"] = 40.5
    run_case("D", hd, 1.004, [40, 40], [235.0, 258.0], [-6.0, 18.0], 2, make_image(rng, 48, 80, 0.02), out)
    # E: fp32-exact image values (as real L2 FITS data, BITPIX=-32, cast to float64)
    img32 = make_image(rng, 64, 64, 0.01).astype(np.float32).astype(np.float64)
    run_case("E", dict(base), 1.004, [64, 64], [230.0, 260.0], [-10.0, 20.0], 2, img32, out)

    # direct interpol2d edge/NaN cases through the reference's rectify.interpol2d (scipy 1.7.1)
    small = make_image(rng, 9, 11, 0.0)
    small[4, 5] = np.nan
    xs = np.array([-0.6, -0.5, -1e-9, 0.0, 0.25, 0.5, 1.49, 1.5, 2.5, 4.5, 5.0, 5.5, 6.49, 6.51, 9.5, 9.75, 10.0,
                   10.0 + 1e-9, 10.4, np.nan, 3.3])
    ys = np.array([0.0, 0.3, 0.49, 0.5, 7.5, 7.6, 8.0, 8.0 + 1e-9, -0.2, np.nan, 3.5, 4.0, 2.51, 5.49])
    gx, gy = np.meshgrid(xs, ys)
    for order in (1, 2):
        res = rectify.interpol2d(small, gx, gy, order=order, fill=-32762)
        out["edge/res_order%d" % order] = np.where(res == -32762, np.nan, res)
        d32 = np.zeros(gx.shape, dtype=np.float32)
        rectify.interpol2d(small, gx, gy, order=order, fill=np.nan, dst=d32)
        out["edge/res32_order%d" % order] = d32
    out["edge/image"] = small
    out["edge/x"] = gx
    out["edge/y"] = gy

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rectify_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes; numpy", np.__version__, "python", sys.version.split()[0])


if __name__ == "__main__":
    main()
