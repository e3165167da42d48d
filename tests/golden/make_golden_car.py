#!/opt/conda/bin/python3.9
"""
Golden vectors for the plate-carree (CAR) WCS arithmetic of the `align_using_initial_carrington` path
(astropy.wcs.WCS -> wcslib; call sites alignment.py:1038-1069 with lon_ctype 'CRLN-CAR', Util.py:282-312): inputs that
are already Carrington maps.  A CRVAL2 lag moves the reference latitude off the equator, which makes the projection an
OBLIQUE plate carree (wcslib celset: native pole at latitude 90 - |CRVAL2|, LONPOLE default 0 or 180 deg by the sign of
CRVAL2) -- the cases below cover both signs, a roll and unequal CDELT.

Run (build container only; astropy 4.3.1 / wcslib 7.6 live under the side interpreter):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_car.py
"""
import os

import numpy as np

for _n, _v in [("asscalar", lambda a: a.item()), ("alen", len)]:
    if not hasattr(np, _n):
        setattr(np, _n, _v)

from astropy.wcs import WCS  # noqa: E402

KEYS = ["NAXIS1", "NAXIS2", "CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "PC1_1", "PC1_2", "PC2_1",
        "PC2_2", "LONPOLE"]


def header(naxis1, naxis2, crpix1, crpix2, crval1, crval2, cdelt1, cdelt2, crota_deg, lonpole=None):
    rho = np.deg2rad(crota_deg)
    lam = cdelt2 / cdelt1
    h = {"NAXIS": 2, "NAXIS1": naxis1, "NAXIS2": naxis2, "CTYPE1": "CRLN-CAR", "CTYPE2": "CRLT-CAR", "CUNIT1": "deg",
         "CUNIT2": "deg", "CRPIX1": crpix1, "CRPIX2": crpix2, "CRVAL1": crval1, "CRVAL2": crval2, "CDELT1": cdelt1,
         "CDELT2": cdelt2, "PC1_1": np.cos(rho), "PC2_2": np.cos(rho), "PC1_2": -lam * np.sin(rho),
         "PC2_1": np.sin(rho) / lam}
    if lonpole is not None:
        h["LONPOLE"] = lonpole
    return h


def store(out, name, h):
    out[name + "/keys"] = np.array([k for k in KEYS if k in h])
    out[name + "/vals"] = np.array([float(h[k]) for k in KEYS if k in h])


def main():
    out = {}
    rng = np.random.default_rng(11)
    hdrs = {
        "equator": header(400, 300, 200.5, 150.5, 250.0, 0.0, 0.05, 0.05, 0.0),
        "north": header(400, 300, 200.5, 150.5, 250.0, 0.0125, 0.05, 0.05, 0.0),
        "south": header(400, 300, 180.0, 140.0, 249.99, -0.02, 0.05, 0.04, 0.0),
        "rolled": header(300, 300, 150.5, 150.5, 120.0, 0.004, 0.03, 0.03, 0.4),
        "high": header(256, 256, 128.5, 128.5, 30.0, 35.0, 0.1, 0.1, 0.0),
        "lonpole_bad": header(256, 256, 128.5, 128.5, 300.0, -5.0, 0.1, 0.1, 0.0, lonpole=0.0),
        "lonpole_ok": header(256, 256, 128.5, 128.5, 300.0, 5.0, 0.1, 0.1, 0.0, lonpole=0.0),
        "lonpole_180": header(256, 256, 128.5, 128.5, 300.0, -5.0, 0.1, 0.1, 0.0, lonpole=180.0),
    }
    invalid = []
    for name, h in list(hdrs.items()):
        try:
            w = WCS(h)
        except Exception as e:  # wcslib celset: no valid native-pole latitude (explicit LONPOLE on the wrong side)
            invalid.append(name)
            store(out, name, h)
            out[name + "/error"] = np.array(type(e).__name__)
            del hdrs[name]
            continue
        px = rng.uniform(-20, h["NAXIS1"] + 20, 400)
        py = rng.uniform(-20, h["NAXIS2"] + 20, 400)
        lon, lat = w.wcs_pix2world(px, py, 0)
        bx, by = w.wcs_world2pix(lon, lat, 0)
        store(out, name, h)
        out[name + "/px"], out[name + "/py"] = px, py
        out[name + "/lon"], out[name + "/lat"] = np.asarray(lon), np.asarray(lat)
        out[name + "/back_x"], out[name + "/back_y"] = np.asarray(bx), np.asarray(by)
        out[name + "/lonpole_used"] = np.array(w.wcs.lonpole)
        out[name + "/latpole_used"] = np.array(w.wcs.latpole)
    # composites: pixel grid of map A -> world -> pixels of (lag-shifted) map B: one lag-point of the sweep
    comps = {
        "lag_ns": (hdrs["equator"], header(380, 280, 190.5, 140.5, 250.0 + 0.03, -0.0175, 0.05, 0.05, 0.0)),
        "lag_nn": (hdrs["north"], header(380, 280, 190.5, 140.5, 250.0 - 0.04, 0.03, 0.05, 0.05, 0.0)),
        "lag_roll": (hdrs["equator"], header(380, 280, 190.5, 140.5, 250.01, 0.01, 0.05, 0.05, 0.3)),
        "lag_cdelt": (hdrs["south"], header(380, 280, 190.5, 140.5, 250.0, -0.01, 0.0502, 0.0398, -0.2)),
    }
    for tag, (ha, hb) in comps.items():
        wa, wb = WCS(ha), WCS(hb)
        gx, gy = np.meshgrid(np.arange(0, ha["NAXIS1"], 13.0), np.arange(0, ha["NAXIS2"], 11.0))
        lon, lat = wa.wcs_pix2world(gx.ravel(), gy.ravel(), 0)
        x, y = wb.wcs_world2pix(lon, lat, 0)
        store(out, tag + "/A", ha)
        store(out, tag + "/B", hb)
        out[tag + "/gx"], out[tag + "/gy"] = gx.ravel(), gy.ravel()
        out[tag + "/x"], out[tag + "/y"] = np.asarray(x), np.asarray(y)
    out["invalid"] = np.array(invalid)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "car_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "astropy", __import__("astropy").__version__)
    for name in hdrs:
        print(name, "lonpole", float(out[name + "/lonpole_used"]), "latpole", float(out[name + "/latpole_used"]),
              "lon[0..2]", out[name + "/lon"][:2], "lat", out[name + "/lat"][:2])


if __name__ == "__main__":
    main()
