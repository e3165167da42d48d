"""
SPICE L2 / L3 header handling for `AlignmentSpice` without astropy (the reference goes through astropy.wcs / wcslib:
hdrshift/alignment_spice.py:250-355, utils/Util.py:429-455).

`celestial_header` restates what `WCS(hdr).dropaxis(...).dropaxis(...).to_header()` leaves of a 4-D SPICE header:
the two helioprojective axes renumbered 1, 2, angular values normalised to degrees and printed with 14 significant
digits (wcslib's WCSHDO_P14, which astropy's to_header uses), PCi_j written only where it differs from the identity,
LONPOLE / LATPOLE defaults and the observer keywords wcslib carries along.  Pinned against astropy 4.3.1 / wcslib 7.6
by tests/golden/spice_header_golden.json.
"""
from __future__ import annotations

import datetime as _dt

import numpy as np

from .fits_io import Header
from .header import unit_to_deg


def _p14(v: float) -> float:
    return float("%.14G" % float(v))


def _axes(hdr):
    naxis = int(hdr.get("WCSAXES", hdr.get("NAXIS", 0)))
    lon = lat = None
    for i in range(1, naxis + 1):
        ct = str(hdr.get("CTYPE%d" % i, "")).strip()
        if ct.startswith("HPLN"):
            lon = i
        elif ct.startswith("HPLT"):
            lat = i
    if lon is None or lat is None:
        raise ValueError("no HPLN-/HPLT- axes in the SPICE header")
    return lon, lat


def parse_date(date: str) -> _dt.datetime:
    s = str(date).strip().rstrip("Z")
    fmt = "%Y-%m-%dT%H:%M:%S.%f" if "." in s else ("%Y-%m-%dT%H:%M:%S" if "T" in s else "%Y-%m-%d")
    return _dt.datetime.strptime(s, fmt)


def _mjd_parts(date: str):
    """(integer MJD, day fraction), the fraction from the time of day itself (no cancellation)."""
    t = parse_date(date)
    days = (t.date() - _dt.date(1858, 11, 17)).days
    frac = (t.hour * 3600 + t.minute * 60 + t.second + t.microsecond * 1e-6) / 86400.0
    return days, frac


def _mjd(date: str) -> float:
    days, frac = _mjd_parts(date)
    return days + frac


def celestial_header(hdr) -> Header:
    """2-D helioprojective header of a SPICE window (alignment_spice.py:255-261 for L2, :350-355 for L3).  The
    coupling of the time axis to the raster direction (PC4_1, zeroed at :257 / :353) disappears with the axis."""
    lon, lat = _axes(hdr)
    ax = (lon, lat)
    out = Header()
    out["WCSAXES"] = 2
    for k, a in enumerate(ax):
        out["CRPIX%d" % (k + 1)] = float(hdr.get("CRPIX%d" % a, 0.0))
    for i, a in enumerate(ax):
        for j, b in enumerate(ax):
            v = float(hdr.get("PC%d_%d" % (a, b), 1.0 if a == b else 0.0))
            if v != (1.0 if i == j else 0.0):
                out["PC%d_%d" % (i + 1, j + 1)] = _p14(v)
    scale = [unit_to_deg(str(hdr.get("CUNIT%d" % a, "deg")).strip() or "deg") for a in ax]
    for k, a in enumerate(ax):
        out["CDELT%d" % (k + 1)] = _p14(float(hdr.get("CDELT%d" % a, 1.0)) * scale[k])
    for k in range(2):
        out["CUNIT%d" % (k + 1)] = "deg"
    for k, a in enumerate(ax):
        out["CTYPE%d" % (k + 1)] = str(hdr["CTYPE%d" % a]).strip()
    for k, a in enumerate(ax):
        out["CRVAL%d" % (k + 1)] = _p14(float(hdr.get("CRVAL%d" % a, 0.0)) * scale[k])
    # zenithal projection: native pole at theta0 = 90 deg, so LONPOLE defaults to 180 deg (delta0 < theta0)
    out["LONPOLE"] = float(hdr.get("LONPOLE", 180.0))
    out["LATPOLE"] = out["CRVAL2"]
    dates = [k for k in ("DATE-OBS", "DATE-BEG", "DATE-AVG", "DATE-END") if k in hdr]
    if "DATEREF" in hdr:
        if "TIMESYS" in hdr:
            out["TIMESYS"] = str(hdr["TIMESYS"]).strip()
        out["DATEREF"] = hdr["DATEREF"]
        days, frac = _mjd_parts(hdr["DATEREF"])
        out["MJDREFI"] = float(days)
        out["MJDREFF"] = _p14(frac)
    elif dates or any(k in hdr for k in ("RSUN_REF", "DSUN_OBS")):
        out["MJDREF"] = 0.0
    for k in dates:
        out[k] = hdr[k]
        out["MJD-" + k[5:]] = _p14(_mjd(hdr[k]))
    for k in ("RSUN_REF", "DSUN_OBS", "CRLN_OBS", "CRLT_OBS", "HGLN_OBS", "HGLT_OBS"):
        if k in hdr:
            out[k] = float(hdr[k])
    return out


def wavelengths_angstrom(hdr) -> np.ndarray:
    """Wavelength of every spectral pixel (alignment_spice.py:258, :274-275), in angstrom."""
    naxis = int(hdr.get("NAXIS", 0))
    for i in range(1, naxis + 1):
        if str(hdr.get("CTYPE%d" % i, "")).strip().startswith("WAVE"):
            unit = str(hdr.get("CUNIT%d" % i, "m")).strip()
            to_a = {"m": 1e10, "nm": 10.0, "angstrom": 1.0, "Angstrom": 1.0, "um": 1e4, "mm": 1e7, "cm": 1e8}[unit]
            z = np.arange(int(hdr["NAXIS%d" % i]), dtype=np.float64)
            pc = float(hdr.get("PC%d_%d" % (i, i), 1.0))
            return (float(hdr["CRVAL%d" % i]) + float(hdr["CDELT%d" % i]) * pc * (z + 1.0 - float(hdr["CRPIX%d" % i]))) * to_a
    raise ValueError("no WAVE axis in the SPICE header")


def column_times(hdr):
    """Time of every raster column [s after the reference epoch] and that epoch: the TIME axis of the (x, y, t) WCS
    the synthetic-raster builder evaluates (synras/map_builder.py:247-288), averaged over the slit as
    `_return_mean_time` does (:237-243).  world_t = CRVAL4 + CDELT4 * sum_j PC4_j (p_j - CRPIX_j)."""
    naxis = int(hdr.get("NAXIS", 0))
    lon, lat = _axes(hdr)
    tax = None
    for i in range(1, naxis + 1):
        if str(hdr.get("CTYPE%d" % i, "")).strip() in ("TIME", "UTC"):
            tax = i
    if tax is None:
        raise ValueError("no TIME axis in the SPICE header")
    nx, ny = int(hdr["NAXIS%d" % lon]), int(hdr["NAXIS%d" % lat])
    x = np.arange(nx, dtype=np.float64) + 1.0 - float(hdr.get("CRPIX%d" % lon, 0.0))
    y = np.arange(ny, dtype=np.float64) + 1.0 - float(hdr.get("CRPIX%d" % lat, 0.0))
    pc_x = float(hdr.get("PC%d_%d" % (tax, lon), 0.0))
    pc_y = float(hdr.get("PC%d_%d" % (tax, lat), 0.0))
    pc_t = float(hdr.get("PC%d_%d" % (tax, tax), 1.0))
    t0 = 1.0 - float(hdr.get("CRPIX%d" % tax, 0.0))  # first (only) time pixel
    cd = float(hdr.get("CDELT%d" % tax, 1.0))
    unit = str(hdr.get("CUNIT%d" % tax, "s")).strip() or "s"
    to_s = {"s": 1.0, "min": 60.0, "h": 3600.0, "d": 86400.0}[unit]
    t = float(hdr.get("CRVAL%d" % tax, 0.0)) + cd * (pc_x * x[None, :] + pc_y * y[:, None] + pc_t * t0)
    ref = None
    for k in ("DATEREF", "DATE-REF", "DATE-BEG", "DATE-OBS"):
        if k in hdr:
            ref = parse_date(hdr[k])
            break
    if ref is None:
        raise ValueError("the SPICE header has no DATEREF / DATE-BEG to anchor its TIME axis")
    return t.mean(axis=0) * to_s, ref


def slit_pxl(header):
    """First and last pixel of the slit (utils/Util.py:431-448)."""
    ybin = header["NBIN2"]
    h_detector = 1024 / ybin
    det = str(header["DETECTOR"]).strip()
    if det == "SW":
        h_slit = 600 / ybin
    elif det == "LW":
        h_slit = 626 / ybin
    else:
        raise ValueError(f"unknown detector: {header['DETECTOR']}")
    slit_beg = (h_detector - h_slit) / 2
    slit_end = h_detector - slit_beg
    slit_beg = slit_beg - header["PXBEG2"] / ybin + 1
    slit_end = slit_end - header["PXBEG2"] / ybin + 1
    return int(np.ceil(slit_beg)), int(np.floor(slit_end))


def vertical_edges_limits(header):
    """utils/Util.py:451-455."""
    iymin, iymax = slit_pxl(header)
    iymin += int(20 / header["NBIN2"])
    iymax -= int(20 / header["NBIN2"])
    return iymin, iymax


def diff_rot(lat, wvl="default"):
    """Differential minus Carrington angular velocity [rad/s] (utils/Util.py:315-345; Hortin 2003 coefficients)."""
    p = {"EIT 171": (14.56, -2.65, 0.96), "EIT 195": (14.50, -2.14, 0.66), "EIT 284": (14.60, -0.71, -1.18),
         "EIT 304": (14.51, -3.12, 0.34)}
    p["default"] = p["EIT 195"]
    A, B, C = p[wvl]
    corr = A - 360 / 25.38 + B * np.sin(lat) ** 2 + C * np.sin(lat) ** 4
    return np.deg2rad(corr / 86400)
