#!/opt/conda/bin/python3.9
"""
Golden vectors for the identity lag of `align_using_initial_carrington` (two CRLN-CAR / CRLT-CAR maps; both branches of
the reference build the sub-map, alignment.py:649-651 / :765-767, so the target header IS the header of the map to align
and the zero lag is pixel -> world -> pixel through ONE header, alignment.py:1038-1069 with lon_ctype "CRLN-CAR": no
ang2pipi).  The round trip returns i + eps, |eps| ~ 1e-13 px, and the sign of eps decides, through the bounds rule of
map_coordinates (utils/Util.py:98-102), whether a border pixel is kept -- wcslib's rounding noise (lin.c, prj.c carx2s /
cars2x, sph.c, cel.c celset for a cylindrical projection), nothing else.

Records wcslib's own numbers (astropy 4.3.1 / wcslib 7.6) for EVERY border pixel plus a diagonal of interior pixels of
several CAR headers: equatorial (the simple-rotation branch of sphx2s), referenced off the equator on either side
(oblique), rolled, with unequal / negative pixel sizes, with an explicit LONPOLE, in arcsec.

Run (build container only):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_border_car.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_border import border, card_value  # noqa: E402  (same card-formatting fixed points)

from astropy.wcs import WCS  # noqa: E402

KEYS = ["NAXIS1", "NAXIS2", "CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "PC1_1", "PC1_2", "PC2_1",
        "PC2_2", "LONPOLE", "LATPOLE"]


def header(naxis1, naxis2, crpix1, crpix2, crval1, crval2, cdelt1, cdelt2, crota_deg, unit="deg", lonpole=None,
           latpole=None):
    rho = np.deg2rad(crota_deg)
    lam = cdelt2 / cdelt1
    h = {"NAXIS": 2, "NAXIS1": naxis1, "NAXIS2": naxis2, "CTYPE1": "CRLN-CAR", "CTYPE2": "CRLT-CAR", "CUNIT1": unit,
         "CUNIT2": unit, "CRPIX1": crpix1, "CRPIX2": crpix2, "CRVAL1": crval1, "CRVAL2": crval2, "CDELT1": cdelt1,
         "CDELT2": cdelt2, "PC1_1": np.cos(rho), "PC2_2": np.cos(rho), "PC1_2": -lam * np.sin(rho),
         "PC2_1": np.sin(rho) / lam}
    if lonpole is not None:
        h["LONPOLE"] = lonpole
    if latpole is not None:
        h["LATPOLE"] = latpole
    for k, v in list(h.items()):
        if isinstance(v, float):
            h[k] = card_value(float(v))
    return h


def main():
    out = {}
    hdrs = {
        # tests' make_car_scene map to align (scene F of alignment_golden): referenced just north of the equator
        "sceneF": header(80, 60, 40.5, 30.5, 250.0533, 0.01137, 0.0101, 0.0099, 0.0),
        # equatorial synoptic-like map: sphx2s / sphs2x take the "simple change in origin of longitude" branch
        "equatorial": header(120, 90, 60.5, 45.5, 180.0, 0.0, 0.5, 0.5, 0.0),
        # south of the equator (native pole on the other side: LONPOLE defaults to 180), rolled, unequal pixels
        "south_rolled": header(97, 71, 40.0, 35.5, 33.25, -12.5, 0.0111, 0.0093, 0.4),
        # negative CDELT1 (longitude running right to left), far north
        "flipped_north": header(64, 48, 32.5, 24.5, 310.5, 41.0, -0.02, 0.02, -2.5),
        # explicit LONPOLE / LATPOLE as astropy's to_header writes them for a northern map
        "explicit_pole": header(80, 60, 40.5, 30.5, 250.0, 0.5, 0.0101, 0.0099, 0.0, lonpole=0.0, latpole=90.0),
        # arcsec units (wcslib's unit fix scales CRVAL / CDELT)
        "arcsec": header(50, 40, 25.5, 20.5, 900.0 * 3600.0 / 10.0, 36.0, 36.36, 35.64, 0.0, unit="arcsec"),
        # two more southern maps (LONPOLE defaults to 180: the general branch of celset), one with LONPOLE spelled out
        "south_plain": header(60, 44, 30.5, 22.5, 250.0, -0.01137, 0.0101, 0.0099, 0.0),
        "south_explicit": header(66, 40, 30.0, 20.5, 120.75, -33.0, 0.03, 0.025, -1.5, lonpole=180.0),
        # negative CRVAL1 (other branch of the longitude normalisation)
        "negative_lon": header(70, 50, 35.5, 25.5, -120.25, 3.0, 0.05, 0.05, 10.0),
    }
    for name, h in hdrs.items():
        w = WCS(h)
        bx, by = border(h["NAXIS1"], h["NAXIS2"])
        n = min(h["NAXIS1"], h["NAXIS2"])
        bx = np.concatenate([bx, np.arange(1, n - 1, dtype=np.float64), np.arange(1, n - 1) + 0.37])
        by = np.concatenate([by, np.arange(1, n - 1, dtype=np.float64), np.arange(1, n - 1) * 0.71 + 0.21])
        lon, lat = w.pixel_to_world_values(bx, by)      # extract_EUI_coordinates, lon_ctype CRLN-CAR: raw values
        rx, ry = w.world_to_pixel_values(lon, lat)      # alignment.py:1065
        out[name + "/keys"] = np.array([k for k in KEYS if k in h])
        out[name + "/vals"] = np.array([float(h[k]) for k in KEYS if k in h])
        out[name + "/unit"] = np.array(h["CUNIT1"])
        out[name + "/bx"], out[name + "/by"] = bx, by
        out[name + "/lon"], out[name + "/lat"] = lon, lat
        out[name + "/rx"], out[name + "/ry"] = rx, ry
        out[name + "/lonpole"] = np.array(w.wcs.lonpole)
        out[name + "/latpole"] = np.array(w.wcs.latpole)
        nx, ny = h["NAXIS1"], h["NAXIS2"]
        nb = 2 * (nx + ny) - 4
        drop = (rx < 0) | (rx > nx - 1) | (ry < 0) | (ry > ny - 1)
        print(name, "pixels", bx.size, "border pixels dropped by the bounds rule", int(drop[:nb].sum()), "of", nb,
              "max |eps|", float(np.nanmax(np.hypot(rx - bx, ry - by))), "lonpole", float(w.wcs.lonpole), "latpole",
              float(w.wcs.latpole))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "border_car_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
