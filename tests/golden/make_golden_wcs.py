#!/opt/conda/bin/python3.9
"""
Golden-vector generator for the third-party TAN WCS arithmetic on the helioprojective path
(astropy.wcs.WCS -> wcslib; call sites alignment.py:1041-1065, Util.py:284-290).

astropy is NOT part of /root/reference (it is a poetry.lock dependency, pinned 7.2.0); the build
container only has astropy 4.3.1 / wcslib 7.6 under the side interpreter.  This script records
`pixel_to_world_values` / `world_to_pixel_values` outputs (the numbers behind the high-level
`pixel_to_world` / `world_to_pixel` calls of the non-sunpy branch) for HRIEUV-, FSI- and SPICE-like
headers, plus the composite "pixel grid of header A -> world -> ang2pipi -> pixel of shifted header B"
that one helioprojective lag-point evaluates.

Run (build container only):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_wcs.py
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
    return h


def ang2pipi(a):
    return -((-a + 180.0) % 360.0 - 180.0)


def store(out, name, h):
    out[name + "/keys"] = np.array([k for k in KEYS if k in h])
    out[name + "/vals"] = np.array([float(h[k]) for k in KEYS if k in h])
    out[name + "/unit"] = np.array(h["CUNIT1"])


def main():
    out = {}
    hdrs = {
        "hri": header(2048, 2048, 1024.5, 1024.5, -310.0, 420.0, 0.492, 0.492, 3.0, "arcsec", lonpole=180.0),
        "fsi": header(3072, 3072, 1536.5, 1536.5, 12.5, -7.25, 4.44, 4.44, 0.0, "arcsec"),
        "spice": header(192, 832, 96.5, 416.5, -0.0861, 0.1166, 4.0 / 3600, 1.098 / 3600, -2.4, "deg"),
        "far": header(512, 512, 200.0, 300.0, 2500.0, -1800.0, 10.0, 10.0, 40.0, "arcsec"),
    }
    rng = np.random.default_rng(7)
    for name, h in hdrs.items():
        w = WCS(h)
        px = np.concatenate([rng.uniform(-5, h["NAXIS1"] + 5, 60), [0.0, h["NAXIS1"] - 1.0, h["CRPIX1"] - 1]])
        py = np.concatenate([rng.uniform(-5, h["NAXIS2"] + 5, 60), [0.0, h["NAXIS2"] - 1.0, h["CRPIX2"] - 1]])
        lon, lat = w.wcs_pix2world(px, py, 0)
        lon2, lat2 = w.pixel_to_world_values(px, py)
        assert np.array_equal(lon, lon2) and np.array_equal(lat, lat2)
        bx, by = w.world_to_pixel_values(ang2pipi(lon), ang2pipi(lat))
        store(out, name, h)
        out[name + "/px"] = px
        out[name + "/py"] = py
        out[name + "/lon"] = lon
        out[name + "/lat"] = lat
        out[name + "/back_x"] = bx
        out[name + "/back_y"] = by
        print(name, "roundtrip err", np.abs(bx - px).max(), np.abs(by - py).max())

    # composite maps (one helioprojective lag-point): pixels of A -> world -> ang2pipi -> pixels of B
    def composite(tag, ha, hb, n=9):
        wa, wb = WCS(ha), WCS(hb)
        gx, gy = np.meshgrid(np.linspace(0, ha["NAXIS1"] - 1, n), np.linspace(0, ha["NAXIS2"] - 1, n))
        lon, lat = wa.pixel_to_world_values(gx, gy)
        x, y = wb.world_to_pixel_values(ang2pipi(lon), ang2pipi(lat))
        store(out, tag + "/A", ha)
        store(out, tag + "/B", hb)
        out[tag + "/gx"] = gx
        out[tag + "/gy"] = gy
        out[tag + "/x"] = x
        out[tag + "/y"] = y
        print(tag, "x range", x.min(), x.max())

    # small grid -> shifted small header (per-lag map, alignment.py:1022)
    hb = header(2048, 2048, 1024.5, 1024.5, -310.0 + 24.0, 420.0 + 6.0, 0.492, 0.492, 3.75, "arcsec", lonpole=180.0)
    composite("lag_hri", hdrs["hri"], hb)
    # small grid -> large header (submap, alignment.py:993)
    composite("sub_hri_fsi", hdrs["hri"], hdrs["fsi"])
    # SPICE-like in degrees with crota + crval lag
    hs = header(192, 832, 96.5, 416.5, -0.0861 - 23.0 / 3600, 0.1166 + 36.0 / 3600, 4.0 / 3600, 1.098 / 3600, -2.4 + 0.7,
                "deg")
    composite("lag_spice", hdrs["spice"], hs)
    # intended CDELT lag semantics (CDELT changed, PC rebuilt)
    hc = header(2048, 2048, 1024.5, 1024.5, -310.0 - 5.0, 420.0 + 2.0, 0.492 + 0.01, 0.492 - 0.02, 3.0, "arcsec",
                lonpole=180.0)
    composite("lag_cdelt", hdrs["hri"], hc)

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wcs_golden.npz")
    np.savez_compressed(dst, **out)
    import astropy
    print("wrote", dst, os.path.getsize(dst), "bytes; astropy", astropy.__version__, "python", sys.version.split()[0])


if __name__ == "__main__":
    main()
