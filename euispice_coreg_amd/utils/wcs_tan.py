"""
2-D gnomonic (TAN) WCS in numpy for the once-per-sweep coordinate bookkeeping of the drop-in API
(astropy.wcs is not a dependency): which small-image pixels fall in a lon/lat box
(hdrshift/alignment.py:863-874) and the regular sub-FOV grid of alignment.py:1082-1127.
Direction-cosine formulation of FITS WCS paper II (pixel -> intermediate -> native unit vector -> rotation).
The per-lag coordinate work of the sweep itself never goes through here: it runs on the GPU.
"""
from __future__ import annotations

import numpy as np

from .header import ang2pipi, unit_to_deg


class TanWcs:
    def __init__(self, hdr):
        u = unit_to_deg(hdr.get("CUNIT1", "deg"))
        self.crpix = np.array([float(hdr["CRPIX1"]), float(hdr["CRPIX2"])])
        self.crval = np.radians([float(hdr["CRVAL1"]) * u, float(hdr["CRVAL2"]) * u])
        cd = np.radians([float(hdr["CDELT1"]) * u, float(hdr["CDELT2"]) * u])
        pc = np.array([[float(hdr.get("PC1_1", 1.0)), float(hdr.get("PC1_2", 0.0))],
                       [float(hdr.get("PC2_1", 0.0)), float(hdr.get("PC2_2", 1.0))]])
        self.m = cd[:, None] * pc  # pixel offset -> intermediate world coordinates (radians)
        self.minv = np.linalg.inv(self.m)
        ap, dp, pp = self.crval[0], self.crval[1], np.radians(float(hdr.get("LONPOLE", 180.0)))
        rz1 = np.array([[np.cos(ap), -np.sin(ap), 0], [np.sin(ap), np.cos(ap), 0], [0, 0, 1]])
        t = np.array([[-np.sin(dp), 0, np.cos(dp)], [0, -1, 0], [np.cos(dp), 0, np.sin(dp)]])
        rz2 = np.array([[np.cos(pp), np.sin(pp), 0], [-np.sin(pp), np.cos(pp), 0], [0, 0, 1]])
        self.rot = rz1 @ t @ rz2  # native -> celestial
        n1 = hdr["ZNAXIS1"] if "ZNAXIS1" in hdr else hdr.get("NAXIS1")
        n2 = hdr["ZNAXIS2"] if "ZNAXIS2" in hdr else hdr.get("NAXIS2")
        self.naxis = (int(n1), int(n2)) if n1 is not None else None

    def pixel_to_world(self, px, py):
        """0-based pixels -> (lon, lat) in degrees, lon in ]-180, 180]."""
        q1 = np.asarray(px, dtype=np.float64) + 1.0 - self.crpix[0]
        q2 = np.asarray(py, dtype=np.float64) + 1.0 - self.crpix[1]
        x = self.m[0, 0] * q1 + self.m[0, 1] * q2
        y = self.m[1, 0] * q1 + self.m[1, 1] * q2
        n = np.stack([-y, x, np.ones_like(x)])
        c = np.tensordot(self.rot, n, axes=1)
        lon = np.degrees(np.arctan2(c[1], c[0]))
        lat = np.degrees(np.arctan2(c[2], np.hypot(c[0], c[1])))
        return ang2pipi(lon), lat

    def world_to_pixel(self, lon, lat):
        lo, la = np.radians(lon), np.radians(lat)
        c = np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)])
        n = np.tensordot(self.rot.T, c, axes=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            xr, yr = n[1] / n[2], -n[0] / n[2]
        q1 = self.minv[0, 0] * xr + self.minv[0, 1] * yr
        q2 = self.minv[1, 0] * xr + self.minv[1, 1] * yr
        bad = ~(n[2] > 0)
        return np.where(bad, np.nan, q1 + self.crpix[0] - 1.0), np.where(bad, np.nan, q2 + self.crpix[1] - 1.0)


def pixel_lonlat(hdr):
    """Longitude / latitude (degrees) of every pixel: AlignEUIUtil.extract_EUI_coordinates (utils/Util.py:282-312),
    non-sunpy branch."""
    w = TanWcs(hdr)
    x, y = np.meshgrid(np.arange(w.naxis[0]), np.arange(w.naxis[1]))
    lon, lat = w.pixel_to_world(x, y)
    return lon, ang2pipi(lat)


def _lims_deg(lims, default_unit):
    """[lo, hi] as degrees; items may be astropy Quantities or plain numbers in `default_unit`."""
    out = []
    for v in lims:
        out.append(float(v.to("deg").value) if hasattr(v, "to") else float(v) * unit_to_deg(default_unit))
    return out


def build_regular_grid(longitude, latitude, lonlims_deg=None, latlims_deg=None):
    """PlotFits.build_regular_grid (utils/Util.py:873-906) for longitude/latitude in degrees."""
    dlon = np.hypot(abs(longitude[0, 1] - longitude[0, 0]), abs(latitude[0, 1] - latitude[0, 0]))
    dlat = np.hypot(abs(longitude[1, 0] - longitude[0, 0]), abs(latitude[1, 0] - latitude[0, 0]))
    lon1d = np.arange(np.min(longitude), np.max(longitude), dlon)
    lat1d = np.arange(np.min(latitude), np.max(latitude), dlat)
    if (lonlims_deg is not None) or (latlims_deg is not None):
        lon1d = lon1d[(lon1d > lonlims_deg[0]) & (lon1d < lonlims_deg[1])]
        lat1d = lat1d[(lat1d > latlims_deg[0]) & (lat1d < latlims_deg[1])]
    long, latg = np.meshgrid(lon1d, lat1d)
    return long, latg, dlon, dlat
