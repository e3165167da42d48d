#!/opt/conda/bin/python3.9
"""
Golden vectors for the ONE lag-point of a helioprojective sweep whose border pixels are decided by wcslib's rounding
noise: the zero lag of the parallelism=True path, where the target header IS the shifted header
(alignment.py:1000 hdr_large := hdr_small.copy(); :1038-1069 pixel -> sky -> ang2pipi -> pixel; bounds rule
utils/Util.py:98-102 -> scipy map_coordinates c < 0 or c > n-1).  For identical headers the round trip returns
i + eps with |eps| ~ 1e-12 px, and the sign of eps decides whether a border pixel is kept.

Records, for several headers, wcslib's own numbers for EVERY border pixel (astropy 4.3.1 / wcslib 7.6, the side
interpreter of the build container): sky coordinates, the ang2pipi'd values and the round-trip pixel coordinates, plus
the linear matrices and Euler angles wcslib derived from the header.

Run (build container only):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_border.py
"""
import os
import sys

import numpy as np

for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)

from astropy.wcs import WCS  # noqa: E402

KEYS = ["NAXIS1", "NAXIS2", "CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "PC1_1", "PC1_2", "PC2_1",
        "PC2_2", "LONPOLE"]


def header(naxis1, naxis2, crpix1, crpix2, crval1, crval2, cdelt1, cdelt2, crota_deg, unit, lonpole=None):
    rho = np.deg2rad(crota_deg)
    lam = cdelt2 / cdelt1
    h = {"NAXIS": 2, "NAXIS1": naxis1, "NAXIS2": naxis2, "CTYPE1": "HPLN-TAN", "CTYPE2": "HPLT-TAN", "CUNIT1": unit,
         "CUNIT2": unit, "CRPIX1": crpix1, "CRPIX2": crpix2, "CRVAL1": crval1, "CRVAL2": crval2, "CDELT1": cdelt1,
         "CDELT2": cdelt2, "PC1_1": np.cos(rho), "PC2_2": np.cos(rho), "PC1_2": -lam * np.sin(rho),
         "PC2_1": np.sin(rho) / lam}
    if lonpole is not None:
        h["LONPOLE"] = lonpole
    for k, v in list(h.items()):
        if isinstance(v, float):
            h[k] = card_value(float(v))
            assert card_value(h[k]) == h[k] and float(str(h[k])) == h[k]
    return h


def card_value(v):
    """The float wcslib parses from the header card astropy 4.3.1 writes for `v` (io/fits/card.py _format_float:
    '%.16G', at most 20 characters).  The reference's pinned astropy (7.2.0) writes str(v), which parses back to `v`
    itself; the headers below are built from values that are fixed points of BOTH, so that the golden numbers do not
    depend on the card formatting."""
    s = f"{v:.16G}"
    if "." not in s and "E" not in s:
        s += ".0"
    if len(s) > 20:
        i = s.find("E")
        s = s[:20] if i < 0 else s[:20 - (len(s) - i)] + s[i:]
    return float(s)


def ang2pipi(a):  # utils/Util.py:76-80 on degree values
    return -((-a + 180.0) % 360.0 - 180.0)


def border(nx, ny):
    """0-based pixel coordinates of the perimeter, row-major order of first appearance."""
    xs = np.concatenate([np.arange(nx), np.arange(nx), np.zeros(ny - 2), np.full(ny - 2, nx - 1.0)])
    ys = np.concatenate([np.zeros(nx), np.full(nx, ny - 1.0), np.arange(1, ny - 1), np.arange(1, ny - 1)])
    return xs.astype(np.float64), ys.astype(np.float64)


def main():
    out = {}
    hdrs = {
        # the headline image to align (euispice_coreg_amd/synthetic.py make_scene: wrong-pointing header, CROTA 3)
        "hri2048": header(2048, 2048, 1024.5, 1024.5, -327.0, 429.0, 0.492, 0.492, 3.0, "arcsec", lonpole=180.0),
        # the 50-pixel images of tests/test_gpu_fuzz.py-like scenes
        "px50": header(50, 50, 25.5, 25.5, -327.0, 429.0, 0.492 * 2048 / 50, 0.492 * 2048 / 50, 3.0, "arcsec",
                       lonpole=180.0),
        # cfg1-like 512^2, positive CRVAL1 (other longitude normalisation branch), no roll
        "hri512": header(512, 512, 256.5, 256.5, 210.0, -95.0, 1.968, 1.968, 0.0, "arcsec"),
        # SPICE-like raster in degrees, CDELT1 != CDELT2, negative roll
        "spice": header(192, 832, 96.5, 416.5, -0.0861, 0.1166, 4.0 / 3600, 1.098 / 3600, -2.4, "deg"),
        # far from disk centre, strong roll, odd sizes
        "far": header(301, 173, 120.0, 99.5, 2500.0, -1800.0, 3.7, 3.1, 40.0, "arcsec"),
    }
    for name, h in hdrs.items():
        w = WCS(h)
        bx, by = border(h["NAXIS1"], h["NAXIS2"])
        lon, lat = w.pixel_to_world_values(bx, by)           # what pixel_to_world's Quantities hold (degrees)
        lon2, lat2 = ang2pipi(lon), ang2pipi(lat)             # Util.py:300-301
        rx, ry = w.world_to_pixel_values(lon2, lat2)         # alignment.py:1065
        out[name + "/keys"] = np.array([k for k in KEYS if k in h])
        out[name + "/vals"] = np.array([float(h[k]) for k in KEYS if k in h])
        out[name + "/unit"] = np.array(h["CUNIT1"])
        out[name + "/bx"] = bx
        out[name + "/by"] = by
        out[name + "/lon"] = lon
        out[name + "/lat"] = lat
        out[name + "/lon_pipi"] = lon2
        out[name + "/lat_pipi"] = lat2
        out[name + "/rx"] = rx
        out[name + "/ry"] = ry
        try:
            out[name + "/piximg"] = np.array(w.wcs.piximg_matrix, dtype=np.float64)
            out[name + "/imgpix"] = np.array(w.wcs.imgpix_matrix, dtype=np.float64)
        except AssertionError:  # unit PC matrix: wcslib keeps no matrices (lin.unity)
            out[name + "/piximg"] = np.full((2, 2), np.nan)
            out[name + "/imgpix"] = np.full((2, 2), np.nan)
        out[name + "/crval_deg"] = np.array(w.wcs.crval, dtype=np.float64)
        out[name + "/cdelt_deg"] = np.array(w.wcs.cdelt, dtype=np.float64)
        out[name + "/lonpole"] = np.array(w.wcs.lonpole)
        out[name + "/latpole"] = np.array(w.wcs.latpole)
        nx, ny = h["NAXIS1"], h["NAXIS2"]
        assert out[name + "/unit"] != "arcsec" or np.array_equal(out[name + "/cdelt_deg"],
                                                                   [h["CDELT1"] * (1.0 / 3600.0), h["CDELT2"] * (1.0 / 3600.0)])
        drop = (rx < 0) | (rx > nx - 1) | (ry < 0) | (ry > ny - 1) | ~np.isfinite(rx) | ~np.isfinite(ry)
        out[name + "/dropped"] = drop
        print(name, "border pixels", bx.size, "dropped by the bounds rule", int(drop.sum()),
              "max |eps|", float(np.nanmax(np.hypot(rx - bx, ry - by))))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "border_golden.npz")
    np.savez_compressed(dst, **out)
    import astropy
    print("wrote", dst, os.path.getsize(dst), "bytes; astropy", astropy.__version__, "python", sys.version.split()[0])


if __name__ == "__main__":
    main()
