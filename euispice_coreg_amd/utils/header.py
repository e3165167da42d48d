"""
Header arithmetic of the alignment path (host side, numpy only).

Mirrors, for the drop-in API:
  hdrshift/alignment.py:580-611   Alignment._check_ant_create_pcij_matrix
  hdrshift/alignment.py:799-842   Alignment._set_initial_header_values (unit handling)
  hdrshift/alignment.py:876-887   Alignment._set_threshold_minmax_to_nan
  utils/Util.py:76-80             AlignCommonUtil.ang2pipi
  utils/Util.py:163-245           AlignCommonUtil.correct_pointing_header / _check_and_create_pcij_crota_hdr
"""
from __future__ import annotations

import warnings

import numpy as np

_UNIT_TO_DEG = {"deg": 1.0, "arcsec": 1.0 / 3600.0, "arcmin": 1.0 / 60.0, "rad": 180.0 / np.pi, "mas": 1.0 / 3.6e6}


def unit_to_deg(unit) -> float:
    u = str(unit).strip()
    if u not in _UNIT_TO_DEG:
        raise ValueError(f"unsupported angular unit {unit!r}")
    return _UNIT_TO_DEG[u]


def convert(value, unit_from, unit_to):
    return np.asarray(value, dtype=np.float64) * (unit_to_deg(unit_from) / unit_to_deg(unit_to))


def ang2pipi(value, unit="deg"):
    """Put angles (expressed in `unit`) in ]-180 deg, +180 deg].  utils/Util.py:76-80."""
    pi = 180.0 / unit_to_deg(unit)
    value = np.asarray(value, dtype=np.float64)
    return -((-value + pi) % (2 * pi) - pi)


def check_and_create_pcij_matrix(hdr, force_crota_0=False, warn=True):
    """alignment.py:580-611 (in place)."""
    if "PC1_1" not in hdr:
        if warn:
            warnings.warn("PCi_j matrix not found in header of the FITS file to align. Adding it to the header.")
        if "CROTA" in hdr:
            crot = hdr["CROTA"]
        elif "CROTA2" in hdr:
            crot = hdr["CROTA2"]
        else:
            if force_crota_0:
                crot = 0.0
                hdr["CROTA"] = 0.0
            else:
                raise ValueError("No, CROTA, CROTA2 or PCi_j matrix in your FITS file. If want to force a CROTA=0, "
                                 "please set the force_crota_0 to True when initializing Alignment ")
        rho = np.deg2rad(crot)
        lam = hdr["CDELT2"] / hdr["CDELT1"]
        hdr["PC1_1"] = np.cos(rho)
        hdr["PC2_2"] = np.cos(rho)
        hdr["PC1_2"] = -lam * np.sin(rho)
        hdr["PC2_1"] = (1 / lam) * np.sin(rho)
    if hdr["PC1_1"] >= 1.0:
        if warn:
            warnings.warn(f'{hdr["PC1_1"]=}, setting to  1.0.')
        hdr["PC1_1"] = 1.0
        hdr["PC2_2"] = 1.0
        hdr["PC1_2"] = 0.0
        hdr["PC2_1"] = 0.0
        hdr["CROTA"] = 0.0
    if "CROTA" not in hdr:
        s = -np.sign(hdr["PC1_2"]) + (hdr["PC1_2"] == 0)
        hdr["CROTA"] = s * np.rad2deg(np.arccos(hdr["PC1_1"]))


def set_threshold_minmax_to_nan(data, vmin=None, vmax=None):
    """alignment.py:876-887 (in place): |v| < vmin or |v| > vmax -> NaN."""
    keep = np.ones(data.shape, dtype=bool)
    with np.errstate(invalid="ignore"):
        if vmin is not None:
            keep[np.abs(data) < vmin] = False
        if vmax is not None:
            keep[np.abs(data) > vmax] = False
    data[~keep] = np.nan


def correct_pointing_header(header, lag_cdelt1, lag_cdelt2, lag_crota, lag_crval1, lag_crval2):
    """utils/Util.py:163-215 (in place): apply a pointing correction (lags in arcsec, crota in deg)."""
    # _check_and_create_pcij_crota_hdr, Util.py:217-245
    if "PC1_1" not in header:
        if "CROTA" in header:
            crot = header["CROTA"]
        elif "CROTA2" in header:
            crot = header["CROTA2"]
        else:
            header["CROTA"] = 0.0
            crot = 0.0
        rho = np.deg2rad(crot)
        lam = header["CDELT2"] / header["CDELT1"]
        header["PC1_1"] = np.cos(rho)
        header["PC2_2"] = np.cos(rho)
        header["PC1_2"] = -lam * np.sin(rho)
        header["PC2_1"] = (1 / lam) * np.sin(rho)
    # `>= 1.0` is the helper's test (Util.py:238); correct_pointing_header's own `> 1.0` (Util.py:166) runs after it
    # and can then never be true
    if header["PC1_1"] >= 1.0:
        header["PC1_1"] = 1.0
        header["PC2_2"] = 1.0
        header["PC1_2"] = 0.0
        header["PC2_1"] = 0.0
        header["CROTA"] = 0.0
    if "CROTA" not in header:
        s = -np.sign(header["PC1_2"]) + (header["PC1_2"] == 0)
        header["CROTA"] = s * np.rad2deg(np.arccos(header["PC1_1"]))
    change_pcij = False
    if lag_crval1 is not None:
        header["CRVAL1"] = header["CRVAL1"] + float(convert(lag_crval1, "arcsec", header["CUNIT1"]))
    if lag_crval2 is not None:
        header["CRVAL2"] = header["CRVAL2"] + float(convert(lag_crval2, "arcsec", header["CUNIT2"]))
    key_rota = None
    if "CROTA" in header:
        key_rota = "CROTA"
        crota = header[key_rota]
    elif "CROTA2" in header:
        key_rota = "CROTA2"
        crota = header[key_rota]
    else:
        crota = np.rad2deg(np.arccos(header["PC1_1"]))
        s = -np.sign(header["PC1_2"]) + (header["PC1_2"] == 0.0)
        crota = crota * s
    if lag_crota is not None:
        crota += lag_crota
        if key_rota is not None:
            header[key_rota] = crota
        change_pcij = True
    if lag_cdelt1 is not None:
        header["CDELT1"] = header["CDELT1"] + float(convert(lag_cdelt1, "arcsec", header["CUNIT1"]))
        change_pcij = True
    if lag_cdelt2 is not None:
        header["CDELT2"] = header["CDELT2"] + float(convert(lag_cdelt2, "arcsec", header["CUNIT2"]))
        change_pcij = True
    if change_pcij:
        theta = np.deg2rad(crota)
        lam = header["CDELT2"] / header["CDELT1"]
        header["PC1_1"] = np.cos(theta)
        header["PC2_2"] = np.cos(theta)
        header["PC1_2"] = -lam * np.sin(theta)
        header["PC2_1"] = (1 / lam) * np.sin(theta)
