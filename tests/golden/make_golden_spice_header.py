#!/opt/conda/bin/python3.9
"""
Golden vectors for the SPICE L2 / L3 header flattening (third-party arithmetic: astropy.wcs.WCS.dropaxis / sub /
to_header -> wcslib; call sites hdrshift/alignment_spice.py:255-261 (L2), :350-355 (L3), :258 + :275 (wavelengths)).

Synthetic SPICE-like 4-D headers (x, y, wavelength, time / coefficient, x, y, time) go through the same sequence of
astropy calls the reference makes; the resulting 2-D header cards and the wavelength axis are stored as JSON.

Run (build container only; astropy 4.3.1 / wcslib 7.6 live under the side interpreter):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_spice_header.py
"""
import json
import os

import numpy as np

for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)

from astropy.io import fits  # noqa: E402
from astropy.wcs import WCS  # noqa: E402


def l2_header(crval1, crval2, crota, cdelt1=4.0, cdelt2=1.098, nx=64, ny=96, nw=20, pc4_1=-60.3, dateref=True):
    rho = np.deg2rad(crota)
    lam = cdelt2 / cdelt1
    h = fits.Header()
    cards = [
        ("NAXIS", 4), ("NAXIS1", nx), ("NAXIS2", ny), ("NAXIS3", nw), ("NAXIS4", 1),
        ("CTYPE1", "HPLN-TAN"), ("CTYPE2", "HPLT-TAN"), ("CTYPE3", "WAVE"), ("CTYPE4", "TIME"),
        ("CUNIT1", "arcsec"), ("CUNIT2", "arcsec"), ("CUNIT3", "nm"), ("CUNIT4", "s"),
        ("CRPIX1", (nx + 1) / 2.0), ("CRPIX2", (ny + 1) / 2.0), ("CRPIX3", (nw + 1) / 2.0), ("CRPIX4", 1.0),
        ("CRVAL1", crval1), ("CRVAL2", crval2), ("CRVAL3", 97.7031), ("CRVAL4", 1975.3),
        ("CDELT1", cdelt1), ("CDELT2", cdelt2), ("CDELT3", 0.00973), ("CDELT4", 1.0),
        ("PC1_1", np.cos(rho)), ("PC1_2", -lam * np.sin(rho)), ("PC2_1", np.sin(rho) / lam), ("PC2_2", np.cos(rho)),
        ("PC3_3", 1.0), ("PC4_4", 1.0), ("PC4_1", pc4_1),
        ("CROTA", crota), ("NBIN2", 1), ("DETECTOR", "SW"), ("PXBEG2", 101),
        ("SOLAR_B0", -3.1), ("RSUN_REF", 695700000.0), ("DSUN_OBS", 5.7e10), ("CRLN_OBS", 250.0),
        ("CRLT_OBS", -3.1), ("HGLN_OBS", 20.0), ("HGLT_OBS", -3.1), ("DATE-AVG", "2022-03-17T00:20:32.100"),
        ("DATE-OBS", "2022-03-17T00:00:32.100"), ("DATE-BEG", "2022-03-17T00:00:32.100"),
    ]
    if dateref:
        cards += [("DATEREF", "2022-03-17T00:00:32.100"), ("TIMESYS", "UTC")]
    for k, v in cards:
        h[k] = v
    return h


def l3_header(crval1, crval2, crota, ncoef=4, nx=48, ny=80):
    rho = np.deg2rad(crota)
    cdelt1, cdelt2 = 4.0, 1.098
    lam = cdelt2 / cdelt1
    h = fits.Header()
    cards = [
        ("NAXIS", 4), ("NAXIS1", ncoef), ("NAXIS2", nx), ("NAXIS3", ny), ("NAXIS4", 1),
        ("CTYPE1", "PARAMETER"), ("CTYPE2", "HPLN-TAN"), ("CTYPE3", "HPLT-TAN"), ("CTYPE4", "TIME"),
        ("CUNIT1", ""), ("CUNIT2", "arcsec"), ("CUNIT3", "arcsec"), ("CUNIT4", "s"),
        ("CRPIX1", 1.0), ("CRPIX2", (nx + 1) / 2.0), ("CRPIX3", (ny + 1) / 2.0), ("CRPIX4", 1.0),
        ("CRVAL1", 1.0), ("CRVAL2", crval1), ("CRVAL3", crval2), ("CRVAL4", 975.0),
        ("CDELT1", 1.0), ("CDELT2", cdelt1), ("CDELT3", cdelt2), ("CDELT4", 1.0),
        ("PC1_1", 1.0), ("PC2_2", np.cos(rho)), ("PC2_3", -lam * np.sin(rho)), ("PC3_2", np.sin(rho) / lam),
        ("PC3_3", np.cos(rho)), ("PC4_4", 1.0), ("PC4_2", -20.1),
        ("CROTA", crota), ("NBIN2", 2), ("DETECTOR", "LW"), ("PXBEG2", 51),
        ("SOLAR_B0", 2.0), ("RSUN_REF", 695700000.0), ("DSUN_OBS", 7.1e10),
    ]
    for k, v in cards:
        h[k] = v
    return h


def cards_of(hdr):
    out = {}
    for k in hdr.keys():
        if not k or k in ("COMMENT", "HISTORY"):
            continue
        v = hdr[k]
        out[k] = v if isinstance(v, (str, bool, int)) else float(v)
    return out


def flatten_l2(h):
    """alignment_spice.py:255-261, :258."""
    w_spice = WCS(h.copy())
    w_xyt = w_spice.dropaxis(2)
    w_xyt.wcs.pc[2, 0] = 0
    w_wave = w_spice.sub(["spectral"])
    w_xy = w_xyt.dropaxis(2)
    z = np.arange(h["NAXIS3"])
    wave_m = np.asarray(w_wave.wcs_pix2world(z, 0)[0], dtype=np.float64)  # pixel_to_world(z): SpectralCoord in m
    # synras/map_builder.py:247-288: time of every raster column through the (x, y, t) WCS
    w3 = w_spice.dropaxis(2)
    if "DATEREF" not in h:
        return w_xy.to_header().copy(), wave_m, str(w_wave.wcs.cunit[0]), None, None
    x, y, t = np.meshgrid(np.arange(h["NAXIS1"]), np.arange(h["NAXIS2"]), np.arange(h["NAXIS4"]))
    lon, lat, utc = w3.pixel_to_world(x, y, t)
    from astropy.time import Time
    ref = Time(h["DATEREF"])
    col_seconds = np.array([((utc[:, ii, 0] - ref).to("s").value).mean() for ii in range(h["NAXIS1"])])
    col_iso = [utc[0, ii, 0].isot for ii in (0, h["NAXIS1"] - 1)]
    return w_xy.to_header().copy(), wave_m, str(w_wave.wcs.cunit[0]), col_seconds, col_iso


def flatten_l3(h):
    """alignment_spice.py:350-355."""
    w_spice = WCS(h.copy())
    w_xyt = w_spice.dropaxis(0)
    w_xyt.wcs.pc[2, 0] = 0
    w_xy = w_xyt.dropaxis(2)
    return w_xy.to_header().copy()


def main():
    out = {"astropy": __import__("astropy").__version__, "cases": []}
    for name, args in [("l2_a", (-86.1, 416.6, -2.4)), ("l2_b", (512.25, -300.125, 0.0)),
                       ("l2_c", (-1033.0, 12.0, 12.5))]:
        h = l2_header(*args, dateref=(name != "l2_b"))
        flat, wave_m, wunit, col_s, col_iso = flatten_l2(h)
        out["cases"].append({"name": name, "level": 2, "input": cards_of(h), "flat": cards_of(flat),
                             "wave": wave_m.tolist(), "wave_unit": wunit,
                             "column_seconds": None if col_s is None else col_s.tolist(),
                             "column_isot_first_last": col_iso})
    for name, args in [("l3_a", (-86.1, 416.6, -2.4))]:
        h = l3_header(*args)
        out["cases"].append({"name": name, "level": 3, "input": cards_of(h), "flat": cards_of(flatten_l3(h))})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spice_header_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
    for c in out["cases"]:
        print(c["name"], json.dumps(c["flat"]))


if __name__ == "__main__":
    main()
