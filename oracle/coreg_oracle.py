"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A numpy/scipy restatement of the euispice_coreg `hdrshift.Alignment` correlation-sweep
hot path.  Nothing in the product package (`euispice_coreg_amd/`) imports this module: it
is used only by `tests/`, by `__graft_entry__.smoke()` and by `bench.py`'s `cpu_baseline`
leg, as the checker / reported baseline -- never as the thing shipped or measured as GPU.

Parity status ("pinning"):
  * Carrington per-lag resample (rows a-13, a-14 of SURVEY.md section 8): PINNED against outputs
    of the reference's own `utils/rectify.py` (CarringtonTransform + Rectifier + interpol2d)
    run in the build container; vectors + generating script in `tests/golden/`.
  * TAN pixel<->world (third-party astropy.wcs / wcslib, pinned astropy 7.2.0 in
    poetry.lock, not vendored): restated from FITS WCS Paper II; PINNED by vectors generated
    with astropy 4.3.1 / wcslib 7.6 (`tests/golden/make_golden_wcs.py`).  The zero lag of the sub-map
    path (identical headers: border pixels decided by wcslib's rounding noise) goes through a scalar,
    operation-by-operation restatement of wcslib (`WcslibTan`), PINNED bit for bit by
    `tests/golden/border_golden.npz` (`make_golden_border.py`).
  * `scipy.ndimage.map_coordinates` (third-party, pinned scipy 1.17.1): the oracle calls
    scipy itself (1.15.3 here); a scalar model of its order-1/2 semantics is also restated
    (`spline_sample_model`) and checked against scipy in tests.
  * Pearson (`hdrshift/c_correlate.py:39-72`): restated; numba absent, arithmetic trivial.
  * `AlignmentResults._compute_shift`: pinned by the reference's embedded fixture
    (`hdrshift/test/test_AlignmentResults.py:33-126,172-173`).

Each function cites the reference file:line it follows (paths relative to the reference
repository root, `euispice_coreg/...`).

Headers are plain `dict`s holding the FITS keywords the path reads.
"""
from __future__ import annotations

import copy
import math
import multiprocessing as mp
from multiprocessing import shared_memory

import numpy as np
from scipy.ndimage import map_coordinates

R_SUN_M = 695700000.0  # astropy.constants.R_sun.value (IAU 2015 nominal), rectify.py:405

_UNIT_TO_DEG = {"deg": 1.0, "arcsec": 1.0 / 3600.0, "arcmin": 1.0 / 60.0, "rad": 180.0 / math.pi,
                "mas": 1.0 / 3600.0e3}


def unit_to_deg(unit: str) -> float:
    return _UNIT_TO_DEG[str(unit).strip()]


# --------------------------------------------------------------------------------------
# utils/Util.py:76-80
def ang2pipi(ang_deg):
    """Put an angle (degrees) in ]-180, +180].  Util.py:76-80."""
    ang_deg = np.asarray(ang_deg, dtype=np.float64)
    return -((-ang_deg + 180.0) % 360.0 - 180.0)


def ang2pipi_unit(val, unit):
    """ang2pipi on a value expressed in `unit` (what Quantity arithmetic does): the wrap is at
    180 deg expressed in that unit.  Util.py:76-80 with astropy Quantity semantics."""
    pi = 180.0 / unit_to_deg(unit)
    val = np.asarray(val, dtype=np.float64)
    return -((-val + pi) % (2 * pi) - pi)


# --------------------------------------------------------------------------------------
# utils/Util.py:82-104 == utils/rectify.py:22-56
def interpol2d(image, x, y, fill, order, dst=None):
    """scipy map_coordinates(order, mode='constant', cval=fill, prefilter=False).
    Util.py:82-104 / rectify.py:22-56.  (`x == np.nan` guard there is always False.)"""
    coords = np.stack((np.asarray(y).ravel(), np.asarray(x).ravel()), axis=0)
    if dst is None:
        dst = np.empty(np.shape(x), dtype=image.dtype)
    out = dst.reshape(-1)  # view (dst is contiguous)
    map_coordinates(image, coords, order=order, mode="constant", cval=fill, output=out, prefilter=False)
    return dst


def spline_sample_model(image, x, y, fill, order):
    """Scalar-semantics model (vectorised numpy) of what scipy's NI_GeometricTransform does for
    order 1/2, mode='constant', prefilter=False (scipy/ndimage/src/ni_interpolation.c; SURVEY a-14):
      * coordinate c on an axis of length n: c<0 or c>n-1 or NaN -> whole sample = fill;
      * order 2: start=floor(c+0.5)-1, t=c-floor(c+0.5), w0=.5(.5-t)^2, w1=.75-t^2, w2=1-w0-w1;
        order 1: start=floor(c), t=c-start, w0=1-t, w1=t;
      * taps outside [0,n-1] mirrored about the edge sample;
      * value = sum over taps (row-major) of (v*wy)*wx in float64.
    Used to pin the semantics the HIP kernel implements."""
    image = np.asarray(image, dtype=np.float64)
    ny_, nx_ = image.shape
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    shp = x.shape
    x = x.ravel()
    y = y.ravel()
    with np.errstate(invalid="ignore"):
        inb = (x >= 0) & (x <= nx_ - 1) & (y >= 0) & (y <= ny_ - 1)
    xs = np.where(inb, x, 0.0)
    ys = np.where(inb, y, 0.0)

    def axis_weights(c, n):
        if order == 2:
            f = np.floor(c + 0.5)
            t = c - f
            w1 = 0.75 - t * t
            h = 0.5 - t
            w0 = 0.5 * h * h
            w2 = 1.0 - w0 - w1
            start = f.astype(np.int64) - 1
            ws = [w0, w1, w2]
        elif order == 1:
            f = np.floor(c)
            t = c - f
            start = f.astype(np.int64)
            ws = [1.0 - t, t]
        else:
            raise NotImplementedError(order)
        idx = []
        for k in range(order + 1):
            i = start + k
            if n > 1:
                s2 = 2 * n - 2
                i = np.where(i < 0, -i, i)
                i = np.where(i >= n, s2 - i, i)
            else:
                i = np.zeros_like(i)
            idx.append(i)
        return ws, idx

    wx, ix = axis_weights(xs, nx_)
    wy, iy = axis_weights(ys, ny_)
    acc = np.zeros_like(xs)
    for a in range(order + 1):
        for b in range(order + 1):
            acc = acc + (image[iy[a], ix[b]] * wy[a]) * wx[b]
    out = np.where(inb, acc, fill)
    return out.reshape(shp)


# --------------------------------------------------------------------------------------
# hdrshift/c_correlate.py:39-72
def c_correlate(s_1, s_2, lags=(0,)):
    """Zero-lag Pearson coefficient as the reference's numba routine computes it
    (mean, centred copies, sum of products / sqrt(sum sq * sum sq)).  c_correlate.py:39-72.
    Empty input -> NaN (the reference divides by zero inside numba)."""
    s_1 = np.asarray(s_1, dtype=np.float64)
    s_2 = np.asarray(s_2, dtype=np.float64)
    if s_1.size == 0:
        return np.array([np.nan])
    c1 = s_1 - float(np.mean(s_1))
    c2 = s_2 - float(np.mean(s_2))
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.sum(c1 * c2) / np.sqrt((c1 ** 2).sum() * (c2 ** 2).sum())
    return np.array([r])


# --------------------------------------------------------------------------------------
# astropy.wcs.WCS restricted to 2-D TAN (FITS WCS papers I/II; wcslib prj.c tanx2s/tans2x,
# sph.c sphx2s/sphs2x).  Call sites: alignment.py:1041-1065, Util.py:284-290.
class TanWCS:
    """2-D gnomonic WCS built from a header dict the way `astropy.wcs.WCS(hdr)` sees it on this
    path: PCi_j + CDELT (PC present after alignment.py:580-611), CUNIT converted to degrees
    (wcslib unitfix), LONPOLE default 180 deg."""

    def __init__(self, hdr):
        u1 = unit_to_deg(hdr.get("CUNIT1", "deg"))
        u2 = unit_to_deg(hdr.get("CUNIT2", "deg"))
        self.crpix = (float(hdr["CRPIX1"]), float(hdr["CRPIX2"]))
        self.cdelt = (float(hdr["CDELT1"]) * u1, float(hdr["CDELT2"]) * u2)
        self.crval = (float(hdr["CRVAL1"]) * u1, float(hdr["CRVAL2"]) * u2)
        self.pc = np.array([[float(hdr.get("PC1_1", 1.0)), float(hdr.get("PC1_2", 0.0))],
                            [float(hdr.get("PC2_1", 0.0)), float(hdr.get("PC2_2", 1.0))]])
        self.lonpole = float(hdr.get("LONPOLE", 180.0))
        self.naxis = (int(hdr["NAXIS1"]) if "NAXIS1" in hdr else None,
                      int(hdr["NAXIS2"]) if "NAXIS2" in hdr else None)
        if "ZNAXIS1" in hdr:
            self.naxis = (int(hdr["ZNAXIS1"]), int(hdr["ZNAXIS2"]))

    def pixel_to_world(self, px, py):
        """0-based pixel -> (lon, lat) degrees (longitude normalised as wcslib does)."""
        px = np.asarray(px, dtype=np.float64)
        py = np.asarray(py, dtype=np.float64)
        q1 = px + 1.0 - self.crpix[0]
        q2 = py + 1.0 - self.crpix[1]
        x = self.cdelt[0] * (self.pc[0, 0] * q1 + self.pc[0, 1] * q2)
        y = self.cdelt[1] * (self.pc[1, 0] * q1 + self.pc[1, 1] * q2)
        # tanx2s
        r = np.hypot(x, y)
        phi = np.where(r == 0.0, 0.0, np.degrees(np.arctan2(x, -y)))
        theta = np.degrees(np.arctan2(180.0 / math.pi, r))
        # sphx2s (wcslib sph.c): direction cosines; latitude via asin, or acos(hypot) when |z| > 0.99
        a0, d0 = self.crval
        dphi = np.radians(phi - self.lonpole)
        th = np.radians(theta)
        sd0, cd0 = math.sin(math.radians(d0)), math.cos(math.radians(d0))
        sth, cth = np.sin(th), np.cos(th)
        xx = sth * cd0 - cth * sd0 * np.cos(dphi)
        yy = -cth * np.sin(dphi)
        zz = sth * sd0 + cth * cd0 * np.cos(dphi)
        dlng = np.degrees(np.arctan2(yy, xx))
        lng = a0 + dlng
        lat_a = np.degrees(np.arcsin(np.clip(zz, -1.0, 1.0)))
        lat_b = np.copysign(np.degrees(np.arccos(np.clip(np.hypot(xx, yy), 0.0, 1.0))), zz)
        lat = np.where(np.abs(zz) > 0.99, lat_b, lat_a)
        # wcslib celx2s normalisation: sign of CRVAL1 decides [0,360) or (-360,0]
        if a0 >= 0.0:
            lng = np.where(lng < 0.0, lng + 360.0, lng)
        else:
            lng = np.where(lng > 0.0, lng - 360.0, lng)
        lng = np.where(lng > 360.0, lng - 360.0, lng)
        lng = np.where(lng < -360.0, lng + 360.0, lng)
        return lng, lat

    def world_to_pixel(self, lon, lat):
        """(lon, lat) degrees -> 0-based pixel (x, y)."""
        lon = np.asarray(lon, dtype=np.float64)
        lat = np.asarray(lat, dtype=np.float64)
        a0, d0 = self.crval
        dl = np.radians(lon - a0)
        la = np.radians(lat)
        sd0, cd0 = math.sin(math.radians(d0)), math.cos(math.radians(d0))
        sla, cla = np.sin(la), np.cos(la)
        # sphs2x + tans2x in direction-cosine form (wcslib switches from asin(z) to acos(hypot(x, y)) when
        # |z| > 0.99; r = r0 cos(theta)/sin(theta) is evaluated here from the cosines directly)
        xx = sla * cd0 - cla * sd0 * np.cos(dl)
        yy = -cla * np.sin(dl)
        zz = sla * sd0 + cla * cd0 * np.cos(dl)
        phi = np.radians(self.lonpole) + np.arctan2(yy, xx)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = (180.0 / math.pi) * np.hypot(xx, yy) / zz
        x = r * np.sin(phi)
        y = -r * np.cos(phi)
        bad = ~(zz > 0.0)
        m = np.array([[self.cdelt[0] * self.pc[0, 0], self.cdelt[0] * self.pc[0, 1]],
                      [self.cdelt[1] * self.pc[1, 0], self.cdelt[1] * self.pc[1, 1]]])
        mi = np.linalg.inv(m)
        q1 = mi[0, 0] * x + mi[0, 1] * y
        q2 = mi[1, 0] * x + mi[1, 1] * y
        px = q1 + self.crpix[0] - 1.0
        py = q2 + self.crpix[1] - 1.0
        px = np.where(bad, np.nan, px)
        py = np.where(bad, np.nan, py)
        return px, py


# --------------------------------------------------------------------------------------
# wcslib's own arithmetic, operation by operation, on libm (scalar).  For the ONE lag-point of a helioprojective
# sweep where the numpy restatement above is not enough: the zero lag of the parallelism=True path, where the target
# header IS the shifted header (alignment.py:1000, :1038-1069).  pixel -> sky -> ang2pipi -> pixel is then the identity up
# to wcslib's rounding noise (|eps| ~ 1e-12..1e-9 px), and the SIGN of that noise decides, through the bounds rule of
# map_coordinates (c < 0 or c > n-1, utils/Util.py:98-102), whether a border pixel is kept -- about half of them are
# dropped.  Restated from wcslib 7.x (third-party, bundled with astropy; absent from the reference tree): lin.c
# linp2x / linx2p / matinv, prj.c tanx2s / tans2x, sph.c sphx2s / sphs2x, wcstrig.c, including the macro expansions
# that fix the order of the roundings (`#define D2R PI/180.0`, `#define R2D 180.0/PI`: angle*D2R is (angle*PI)/180,
# atan2(y,x)*R2D is (atan2(y,x)*180)/PI) and glibc's sincos().  PINNED bit for bit, for every border pixel of five
# headers, by tests/golden/border_golden.npz (astropy 4.3.1 / wcslib 7.6).  libm-dependent, as the reference is.
import ctypes  # noqa: E402
import os  # noqa: E402

_libm = ctypes.CDLL("libm.so.6")
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
_libm.sincos.restype = None
def _sincos(x):
    s = ctypes.c_double(); c = ctypes.c_double()
    _libm.sincos(x, ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value
D2R = math.pi / 180.0
R2D = 180.0 / math.pi

def cosd(a):
    if math.fmod(a, 90.0) == 0.0:
        i = abs(int(math.floor(a / 90.0 + 0.5))) % 4
        return (1.0, 0.0, -1.0, 0.0)[i]
    return math.cos(a * math.pi / 180.0)

def sind(a):
    if math.fmod(a, 90.0) == 0.0:
        i = abs(int(math.floor(a / 90.0 - 0.5))) % 4
        return (1.0, 0.0, -1.0, 0.0)[i]
    return math.sin(a * math.pi / 180.0)

def sincosd(a):
    if math.fmod(a, 90.0) == 0.0:
        i = abs(int(math.floor(a / 90.0 + 0.5))) % 4
        if i == 0: return 0.0, 1.0
        if i == 1: return (1.0 if a > 0.0 else -1.0), 0.0
        if i == 2: return 0.0, -1.0
        return (-1.0 if a > 0.0 else 1.0), 0.0
    return _sincos(a * math.pi / 180.0)

def atan2d(y, x):
    if y == 0.0:
        if x >= 0.0: return 0.0
        if x < 0.0: return 180.0
    elif x == 0.0:
        if y > 0.0: return 90.0
        if y < 0.0: return -90.0
    return math.atan2(y, x) * 180.0 / math.pi

def asind(v):
    if v <= -1.0:
        if v + 1.0 > -1e-12: return -90.0   # wcstrig tolerance (WCSTRIG_TOL = 1e-10)
    elif v == 0.0:
        return 0.0
    elif v >= 1.0:
        if v - 1.0 < 1e-12: return 90.0
    return math.asin(v) * 180.0 / math.pi

def acosd(v):
    if v >= 1.0:
        if v - 1.0 < 1e-10: return 0.0
    elif v == 0.0:
        return 90.0
    elif v <= -1.0:
        if v + 1.0 > -1e-10: return 180.0
    return math.acos(v) * 180.0 / math.pi

def matinv2(m):
    """lin.c matinv() for n = 2: LU with scaled partial pivoting, then column-by-column solve."""
    n = 2
    mxl = [0, 1]; lxm = [0, 0]; rowmax = [0.0, 0.0]
    lu = [[m[0][0], m[0][1]], [m[1][0], m[1][1]]]
    for i in range(n):
        for j in range(n):
            d = abs(m[i][j])
            if d > rowmax[i]: rowmax[i] = d
    for k in range(n):
        colmax = abs(lu[k][k]) / rowmax[k]
        pivot = k
        for i in range(k + 1, n):
            d = abs(lu[i][k]) / rowmax[i]
            if d > colmax:
                colmax = d; pivot = i
        if pivot > k:
            lu[pivot], lu[k] = lu[k], lu[pivot]
            rowmax[pivot], rowmax[k] = rowmax[k], rowmax[pivot]
            mxl[pivot], mxl[k] = mxl[k], mxl[pivot]
        for i in range(k + 1, n):
            if lu[i][k] != 0.0:
                lu[i][k] /= lu[k][k]
                for j in range(k + 1, n):
                    lu[i][j] -= lu[i][k] * lu[k][j]
    for i in range(n):
        lxm[mxl[i]] = i
    inv = [[0.0, 0.0], [0.0, 0.0]]
    for k in range(n):
        inv[lxm[k]][k] = 1.0
        for i in range(lxm[k] + 1, n):
            for j in range(lxm[k], i):
                inv[i][k] -= lu[i][j] * inv[j][k]
        for i in range(n - 1, -1, -1):
            for j in range(i + 1, n):
                inv[i][k] -= lu[i][j] * inv[j][k]
            inv[i][k] /= lu[i][i]
    return inv

class WcslibTan:
    def __init__(self, crpix, cdelt_deg, crval_deg, pc, lonpole=180.0):
        self.crpix = crpix
        self.unity = (pc[0][0] == 1.0 and pc[1][1] == 1.0 and pc[0][1] == 0.0 and pc[1][0] == 0.0)
        self.cdelt = cdelt_deg
        self.piximg = [[cdelt_deg[0] * pc[0][0], cdelt_deg[0] * pc[0][1]],
                       [cdelt_deg[1] * pc[1][0], cdelt_deg[1] * pc[1][1]]]
        self.imgpix = None if self.unity else matinv2(self.piximg)
        # celset, zenithal: euler = (lng0, 90 - lat0, phiP, cos, sin)
        self.e0 = crval_deg[0]
        self.e1 = 90.0 - crval_deg[1]
        self.e2 = lonpole
        self.e4, self.e3 = sincosd(self.e1)   # e3 = cos(e1), e4 = sin(e1)

    def p2s(self, px0, py0):
        # astropy: pixcrd += 1 - origin; linp2x
        t0 = (px0 + 1.0) - self.crpix[0]
        t1 = (py0 + 1.0) - self.crpix[1]
        if self.unity:
            x = self.cdelt[0] * t0
            y = self.cdelt[1] * t1
        else:
            x = 0.0; y = 0.0
            x += self.piximg[0][0] * t0
            y += self.piximg[1][0] * t0
            x += self.piximg[0][1] * t1
            y += self.piximg[1][1] * t1
        # tanx2s
        xj = x + 0.0
        yj = y + 0.0
        r = math.sqrt(xj * xj + yj * yj)
        phi = 0.0 if r == 0.0 else atan2d(xj, -yj)
        theta = atan2d(R2D, r)
        # sphx2s
        dphi = phi - self.e2
        sinthe, costhe = sincosd(theta)
        costhe3 = costhe * self.e3; costhe4 = costhe * self.e4
        sinthe3 = sinthe * self.e3; sinthe4 = sinthe * self.e4
        sinphi, cosphi = sincosd(dphi)
        xx = sinthe4 - costhe3 * cosphi
        if abs(xx) < 1e-5:
            xx = -cosd(theta + self.e1) + costhe3 * (1.0 - cosphi)
        yy = -costhe * sinphi
        if xx != 0.0 or yy != 0.0:
            dlng = atan2d(yy, xx)
        else:
            dlng = dphi + 180.0
        lng = self.e0 + dlng
        if self.e0 >= 0.0:
            if lng < 0.0: lng += 360.0
        else:
            if lng > 0.0: lng -= 360.0
        if lng > 360.0: lng -= 360.0
        elif lng < -360.0: lng += 360.0
        if math.fmod(dphi, 180.0) == 0.0:
            lat = theta + cosphi * self.e1
            if lat > 90.0: lat = 180.0 - lat
            if lat < -90.0: lat = -180.0 - lat
        else:
            z = sinthe3 + costhe4 * cosphi
            if abs(z) > 0.99:
                lat = math.copysign(acosd(math.sqrt(xx * xx + yy * yy)), z)
            else:
                lat = asind(z)
        return lng, lat

    def s2p(self, lng, lat):
        dlng = lng - self.e0
        sinlat, coslat = sincosd(lat)
        coslat3 = coslat * self.e3; coslat4 = coslat * self.e4
        sinlat3 = sinlat * self.e3; sinlat4 = sinlat * self.e4
        sinlng, coslng = sincosd(dlng)
        xx = sinlat4 - coslat3 * coslng
        if abs(xx) < 1e-5:
            xx = -cosd(lat + self.e1) + coslat3 * (1.0 - coslng)
        yy = -coslat * sinlng
        if xx != 0.0 or yy != 0.0:
            dphi = atan2d(yy, xx)
        else:
            dphi = dlng - 180.0
        phi = math.fmod(self.e2 + dphi, 360.0)
        if phi > 180.0: phi -= 360.0
        elif phi < -180.0: phi += 360.0
        if math.fmod(dlng, 180.0) == 0.0:
            theta = lat + coslng * self.e1
            if theta > 90.0: theta = 180.0 - theta
            if theta < -90.0: theta = -180.0 - theta
        else:
            z = sinlat3 + coslat4 * coslng
            if abs(z) > 0.99:
                theta = math.copysign(acosd(math.sqrt(xx * xx + yy * yy)), z)
            else:
                theta = asind(z)
        # tans2x
        sinphi, cosphi = sincosd(phi)
        s = sind(theta)
        if s == 0.0:
            return float("nan"), float("nan")
        r = R2D * cosd(theta) / s
        if s < 0.0:
            return float("nan"), float("nan")
        x = r * sinphi - 0.0
        y = -r * cosphi - 0.0
        # linx2p
        if self.unity:
            p0 = x / self.cdelt[0] + self.crpix[0]
            p1 = y / self.cdelt[1] + self.crpix[1]
        else:
            p0 = 0.0
            p0 += self.imgpix[0][0] * x
            p0 += self.imgpix[0][1] * y
            p0 += self.crpix[0]
            p1 = 0.0
            p1 += self.imgpix[1][0] * x
            p1 += self.imgpix[1][1] * y
            p1 += self.crpix[1]
        return p0 - 1.0, p1 - 1.0


    @classmethod
    def from_header(cls, hdr):
        u1 = unit_to_deg(hdr.get("CUNIT1", "deg"))
        u2 = unit_to_deg(hdr.get("CUNIT2", "deg"))
        return cls([float(hdr["CRPIX1"]), float(hdr["CRPIX2"])], [float(hdr["CDELT1"]) * u1, float(hdr["CDELT2"]) * u2],
                   [float(hdr["CRVAL1"]) * u1, float(hdr["CRVAL2"]) * u2],
                   [[float(hdr.get("PC1_1", 1.0)), float(hdr.get("PC1_2", 0.0))],
                    [float(hdr.get("PC2_1", 0.0)), float(hdr.get("PC2_2", 1.0))]], float(hdr.get("LONPOLE", 180.0)))


class WcslibCar(WcslibTan):
    """The same chain for a plate-carree header (CTYPE -CAR: align_using_initial_carrington, alignment.py:344-399):
    wcslib's cel.c celset() for a cylindrical projection (fiducial native point (phi0, theta0) = (0, 0)) in plain double,
    operation by operation; prj.c carx2s / cars2x (with the default r0 the scale factors are exactly 1: phi = x,
    theta = y); sph.c sphx2s / sphs2x INCLUDING their "simple change in origin of longitude" branch, which an
    equatorial map (CRVAL2 = 0: the native pole is the celestial pole) takes.  PINNED bit for bit by
    tests/golden/border_car_golden.npz (astropy 4.3.1 / wcslib 7.6, seven headers)."""

    def __init__(self, crpix, cdelt_deg, crval_deg, pc, lonpole=None, latpole=None):
        super().__init__(crpix, cdelt_deg, crval_deg, pc, 0.0)
        tol = 1.0e-10
        lng0, lat0 = crval_deg
        phi0 = theta0 = 0.0
        latp = 90.0 if latpole is None or latpole != latpole else float(latpole)
        if lonpole is None or lonpole != lonpole or lonpole == 999.0:
            phip = 180.0 if lat0 < theta0 else 0.0
            phip += phi0
            if phip < -180.0: phip += 360.0
            elif phip > 180.0: phip -= 360.0
        else:
            phip = float(lonpole)
        slat0, clat0 = sincosd(lat0)
        sthe0, cthe0 = sincosd(theta0)
        latpreq = 0
        if phip == phi0:
            sphip, cphip = 0.0, 1.0
            u = theta0
            v = 90.0 - lat0
        else:
            # (measured against astropy 4.3.1 / wcslib 7.6 on southern maps, LONPOLE = 180: the sine that enters the
            # longitude of the native pole is libm's sin(pi) = 1.22e-16, not wcstrig's exact 0 -- lngp comes out one ulp
            # below CRVAL1; pinned by three southern headers of border_car_golden.npz)
            sphip, cphip = _sincos((phip - phi0) * math.pi / 180.0)
            x = cthe0 * cphip
            y = sthe0
            z = math.sqrt(x * x + y * y)
            if z == 0.0:
                if slat0 != 0.0:
                    raise InvalidTransformError("Invalid coordinate transformation parameters")
                latpreq = 2
                if latp > 90.0: latp = 90.0
                elif latp < -90.0: latp = -90.0
            else:
                slz = slat0 / z
                if abs(slz) > 1.0:
                    if (abs(slz) - 1.0) < tol:
                        slz = 1.0 if slz > 0.0 else -1.0
                    else:
                        raise InvalidTransformError("Invalid coordinate transformation parameters")
                u = atan2d(y, x)
                v = acosd(slz)
        if latpreq == 0:
            latp1 = u + v
            if latp1 > 180.0: latp1 -= 360.0
            elif latp1 < -180.0: latp1 += 360.0
            latp2 = u - v
            if latp2 > 180.0: latp2 -= 360.0
            elif latp2 < -180.0: latp2 += 360.0
            if abs(latp - latp1) < abs(latp - latp2):
                latp = latp1 if abs(latp1) < 90.0 + tol else latp2
            else:
                latp = latp2 if abs(latp2) < 90.0 + tol else latp1
            if abs(latp) < 90.0 + tol:
                if latp > 90.0: latp = 90.0
                elif latp < -90.0: latp = -90.0
            else:
                raise InvalidTransformError("Invalid coordinate transformation parameters: no valid solution for latp")
        z = cosd(latp) * clat0
        if abs(z) < tol:
            if abs(clat0) < tol:
                lngp = lng0
            elif latp > 0.0:
                lngp = lng0 + phip - phi0 - 180.0
            else:
                lngp = lng0 - phip + phi0
        else:
            x = (sthe0 - sind(latp) * slat0) / z
            y = sphip * cthe0 / clat0
            if x == 0.0 and y == 0.0:
                raise InvalidTransformError("Invalid coordinate transformation parameters")
            lngp = lng0 - atan2d(y, x)
        if lng0 >= 0.0:
            if lngp < 0.0: lngp += 360.0
            elif lngp > 360.0: lngp -= 360.0
        else:
            if lngp > 0.0: lngp -= 360.0
            elif lngp < -360.0: lngp += 360.0
        self.e0 = lngp
        self.e1 = 90.0 - latp
        self.e2 = phip
        self.e4, self.e3 = sincosd(self.e1)
        self.latpole = latp

    def _lin_p2x(self, px0, py0):
        t0 = (px0 + 1.0) - self.crpix[0]
        t1 = (py0 + 1.0) - self.crpix[1]
        if self.unity:
            return self.cdelt[0] * t0, self.cdelt[1] * t1
        x = 0.0; y = 0.0
        x += self.piximg[0][0] * t0
        y += self.piximg[1][0] * t0
        x += self.piximg[0][1] * t1
        y += self.piximg[1][1] * t1
        return x, y

    def p2s(self, px0, py0):
        x, y = self._lin_p2x(px0, py0)
        # carx2s: s = w[1] * (x + x0) with w[1] = 1, x0 = 0
        phi = 1.0 * (x + 0.0)
        theta = 1.0 * (y + 0.0)
        # sphx2s
        if self.e4 == 0.0:
            if self.e1 == 0.0:
                dlng = math.fmod(self.e0 + 180.0 - self.e2, 360.0)
                lng = phi + dlng
                lat = theta
            else:
                dlng = math.fmod(self.e0 + self.e2, 360.0)
                lng = dlng - phi
                lat = -theta
            if self.e0 >= 0.0:
                if lng < 0.0: lng += 360.0
            else:
                if lng > 0.0: lng -= 360.0
            if lng > 360.0: lng -= 360.0
            elif lng < -360.0: lng += 360.0
            return lng, lat
        dphi = phi - self.e2
        sinthe, costhe = sincosd(theta)
        costhe3 = costhe * self.e3; costhe4 = costhe * self.e4
        sinthe3 = sinthe * self.e3; sinthe4 = sinthe * self.e4
        sinphi, cosphi = sincosd(dphi)
        xx = sinthe4 - costhe3 * cosphi
        if abs(xx) < 1e-5:
            xx = -cosd(theta + self.e1) + costhe3 * (1.0 - cosphi)
        yy = -costhe * sinphi
        if xx != 0.0 or yy != 0.0:
            dlng = atan2d(yy, xx)
        else:
            dlng = dphi + 180.0
        lng = self.e0 + dlng
        if self.e0 >= 0.0:
            if lng < 0.0: lng += 360.0
        else:
            if lng > 0.0: lng -= 360.0
        if lng > 360.0: lng -= 360.0
        elif lng < -360.0: lng += 360.0
        if math.fmod(dphi, 180.0) == 0.0:
            lat = theta + cosphi * self.e1
            if lat > 90.0: lat = 180.0 - lat
            if lat < -90.0: lat = -180.0 - lat
        else:
            z = sinthe3 + costhe4 * cosphi
            if abs(z) > 0.99:
                lat = math.copysign(acosd(math.sqrt(xx * xx + yy * yy)), z)
            else:
                lat = asind(z)
        return lng, lat

    def s2p(self, lng, lat):
        if self.e4 == 0.0:
            if self.e1 == 0.0:
                dphi = math.fmod(self.e2 - 180.0 - self.e0, 360.0)
                phi = math.fmod(lng + dphi, 360.0)
                theta = lat
            else:
                dphi = math.fmod(self.e2 + self.e0, 360.0)
                phi = math.fmod(dphi - lng, 360.0)
                theta = -lat
            if phi > 180.0: phi -= 360.0
            elif phi < -180.0: phi += 360.0
        else:
            dlng = lng - self.e0
            sinlat, coslat = sincosd(lat)
            coslat3 = coslat * self.e3; coslat4 = coslat * self.e4
            sinlat3 = sinlat * self.e3; sinlat4 = sinlat * self.e4
            sinlng, coslng = sincosd(dlng)
            xx = sinlat4 - coslat3 * coslng
            if abs(xx) < 1e-5:
                xx = -cosd(lat + self.e1) + coslat3 * (1.0 - coslng)
            yy = -coslat * sinlng
            if xx != 0.0 or yy != 0.0:
                dphi = atan2d(yy, xx)
            else:
                dphi = dlng - 180.0
            phi = math.fmod(self.e2 + dphi, 360.0)
            if phi > 180.0: phi -= 360.0
            elif phi < -180.0: phi += 360.0
            if math.fmod(dlng, 180.0) == 0.0:
                theta = lat + coslng * self.e1
                if theta > 90.0: theta = 180.0 - theta
                if theta < -90.0: theta = -180.0 - theta
            else:
                z = sinlat3 + coslat4 * coslng
                if abs(z) > 0.99:
                    theta = math.copysign(acosd(math.sqrt(xx * xx + yy * yy)), z)
                else:
                    theta = asind(z)
        # cars2x: x = w[0] * phi - x0 with w[0] = 1, x0 = 0
        x = 1.0 * phi - 0.0
        y = 1.0 * theta - 0.0
        if self.unity:
            p0 = x / self.cdelt[0] + self.crpix[0]
            p1 = y / self.cdelt[1] + self.crpix[1]
        else:
            p0 = 0.0
            p0 += self.imgpix[0][0] * x
            p0 += self.imgpix[0][1] * y
            p0 += self.crpix[0]
            p1 = 0.0
            p1 += self.imgpix[1][0] * x
            p1 += self.imgpix[1][1] * y
            p1 += self.crpix[1]
        return p0 - 1.0, p1 - 1.0

    @classmethod
    def from_header(cls, hdr):
        u1 = unit_to_deg(hdr.get("CUNIT1", "deg"))
        u2 = unit_to_deg(hdr.get("CUNIT2", "deg"))
        return cls([float(hdr["CRPIX1"]), float(hdr["CRPIX2"])], [float(hdr["CDELT1"]) * u1, float(hdr["CDELT2"]) * u2],
                   [float(hdr["CRVAL1"]) * u1, float(hdr["CRVAL2"]) * u2],
                   [[float(hdr.get("PC1_1", 1.0)), float(hdr.get("PC1_2", 0.0))],
                    [float(hdr.get("PC2_1", 0.0)), float(hdr.get("PC2_2", 1.0))]],
                   hdr.get("LONPOLE"), hdr.get("LATPOLE"))


# The same chain in plain C (oracle/csrc/wcslib_tan.c, built by oracle/Makefile): identical arithmetic on the same libm,
# checked bit for bit against the Python class above and against the astropy / wcslib golden vectors
# (tests/test_oracle_golden.py).  It exists so that EVERY pixel of a 2048 x 2048 grid can be re-evaluated (odd spline
# orders at noise-decided lag-points); without the built library the scalar Python loop runs instead, however long.
_WCSTAN_LIB = None


def _wcstan_lib():
    global _WCSTAN_LIB
    if _WCSTAN_LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_wcstan.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.oracle_wcstan_pixel_to_pixel.restype = ctypes.c_int
            lib.oracle_wcstan_pixel_to_pixel.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int64] + [ctypes.c_void_p] * 6
            _WCSTAN_LIB = lib
        else:
            _WCSTAN_LIB = False
    return _WCSTAN_LIB or None


def _wcstan_params(hdr):
    u1 = unit_to_deg(hdr.get("CUNIT1", "deg"))
    u2 = unit_to_deg(hdr.get("CUNIT2", "deg"))
    return np.array([float(hdr["CRPIX1"]), float(hdr["CRPIX2"]), float(hdr["CDELT1"]) * u1, float(hdr["CDELT2"]) * u2,
                     float(hdr["CRVAL1"]) * u1, float(hdr["CRVAL2"]) * u2, float(hdr.get("PC1_1", 1.0)),
                     float(hdr.get("PC1_2", 0.0)), float(hdr.get("PC2_1", 0.0)), float(hdr.get("PC2_2", 1.0)),
                     float(hdr.get("LONPOLE", 180.0))], dtype=np.float64)


def wcslib_pixel_to_pixel(hdr_from, hdr_to, px, py, force_python=False):
    """pixel (0-based) of `hdr_from` -> sky -> ang2pipi -> pixel of `hdr_to` through wcslib's own arithmetic, for arrays
    of pixels: the C twin when built, else the scalar Python class.  Returns (x, y, lng_deg, lat_deg)."""
    px = np.ascontiguousarray(px, dtype=np.float64).ravel()
    py = np.ascontiguousarray(py, dtype=np.float64).ravel()
    ox, oy, lng, lat = (np.empty_like(px) for _ in range(4))
    if str(hdr_from.get("CTYPE1", "")).strip().upper().endswith("-CAR"):
        # two Carrington maps (alignment.py:1061-1065 with lon_ctype "CRLN-CAR": the raw wcslib values, no ang2pipi)
        wf, wt = WcslibCar.from_header(hdr_from), WcslibCar.from_header(hdr_to)
        for k in range(px.size):
            lng[k], lat[k] = wf.p2s(float(px[k]), float(py[k]))
            ox[k], oy[k] = wt.s2p(float(lng[k]), float(lat[k]))
        return ox, oy, lng, lat
    lib = None if force_python else _wcstan_lib()
    if lib is not None:
        pf, pt = _wcstan_params(hdr_from), _wcstan_params(hdr_to)
        lib.oracle_wcstan_pixel_to_pixel(pf.ctypes.data, pt.ctypes.data, px.size, px.ctypes.data, py.ctypes.data,
                                         ox.ctypes.data, oy.ctypes.data, lng.ctypes.data, lat.ctypes.data)
        return ox, oy, lng, lat
    wf, wt = WcslibTan.from_header(hdr_from), WcslibTan.from_header(hdr_to)
    for k in range(px.size):
        lng[k], lat[k] = wf.p2s(float(px[k]), float(py[k]))
        ox[k], oy[k] = wt.s2p(float(ang2pipi(np.float64(lng[k]))), float(ang2pipi(np.float64(lat[k]))))
    return ox, oy, lng, lat


def wcslib_refine_near_integers(hdr_from, hdr_to, x, y, tol=1e-6):
    """Odd spline orders take floor(c) as their first tap (scipy ni_interpolation.c), so where the map returns a
    coordinate within `tol` of an INTEGER -- every pixel of a noise-decided lag-point, see `wcslib_refine_near_bounds` --
    the sign of wcslib's rounding noise decides WHICH taps are used, i.e. which neighbour's NaN poisons the sample.
    Same remedy: wcslib's own arithmetic for those coordinates, whatever their number (the C twin of `WcslibTan` takes
    about a second for a 2048 x 2048 grid)."""
    with np.errstate(invalid="ignore"):
        near = (np.abs(x - np.rint(x)) < tol) | (np.abs(y - np.rint(y)) < tol)
    jj, ii = np.nonzero(near)
    if jj.size == 0:
        return 0
    nx, ny, _, _ = wcslib_pixel_to_pixel(hdr_from, hdr_to, ii.astype(np.float64), jj.astype(np.float64))
    x[jj, ii] = nx
    y[jj, ii] = ny
    return int(jj.size)


def wcslib_refine_near_bounds(hdr_from, hdr_to, x, y, tol=1e-6):
    """Re-evaluate, in place, with wcslib's own arithmetic (scalar, libm) every coordinate of the map `hdr_from` pixels ->
    `hdr_to` pixels that lies within `tol` px of the bounds 0 / NAXIS-1 of `hdr_to`: there the sign of wcslib's rounding
    noise decides the bounds rule of map_coordinates, and nothing but wcslib's arithmetic reproduces it.  Structured
    cases (whole border rows / columns): the zero lag of the sub-map path (identity), CDELT-only lags at zero CRVAL lag
    (one axis invariant)."""
    nx, ny = int(hdr_to["NAXIS1"]), int(hdr_to["NAXIS2"])
    with np.errstate(invalid="ignore"):
        near = (np.abs(x) < tol) | (np.abs(x - (nx - 1)) < tol) | (np.abs(y) < tol) | (np.abs(y - (ny - 1)) < tol)
    jj, ii = np.nonzero(near)
    if jj.size == 0:
        return 0
    bx, by, _, _ = wcslib_pixel_to_pixel(hdr_from, hdr_to, ii.astype(np.float64), jj.astype(np.float64))
    x[jj, ii] = bx
    y[jj, ii] = by
    return int(jj.size)


class InvalidTransformError(ValueError):
    """wcslib celset: 'No valid solution for latp' (astropy.wcs raises InvalidTransformError)."""


class CarWCS:
    """2-D plate-carree (CAR) WCS the way `astropy.wcs.WCS(hdr)` / wcslib evaluates it (FITS WCS paper II, wcslib
    cel.c celset + sph.c): native (phi, theta) = intermediate (x, y) in degrees, fiducial point (phi0, theta0) = (0, 0)
    at (CRVAL1, CRVAL2); LONPOLE defaults to 0 deg when CRVAL2 >= 0, else 180 deg; LATPOLE to 90 deg.  With
    CRVAL2 != 0 the projection is oblique: the native pole sits at celestial latitude 90 - |CRVAL2|."""

    def __init__(self, hdr):
        u1 = unit_to_deg(hdr.get("CUNIT1", "deg"))
        u2 = unit_to_deg(hdr.get("CUNIT2", "deg"))
        self.crpix = (float(hdr["CRPIX1"]), float(hdr["CRPIX2"]))
        self.cdelt = (float(hdr["CDELT1"]) * u1, float(hdr["CDELT2"]) * u2)
        self.crval = (float(hdr["CRVAL1"]) * u1, float(hdr["CRVAL2"]) * u2)
        self.pc = np.array([[float(hdr.get("PC1_1", 1.0)), float(hdr.get("PC1_2", 0.0))],
                            [float(hdr.get("PC2_1", 0.0)), float(hdr.get("PC2_2", 1.0))]])
        self.naxis = (int(hdr["NAXIS1"]) if "NAXIS1" in hdr else None,
                      int(hdr["NAXIS2"]) if "NAXIS2" in hdr else None)
        if "ZNAXIS1" in hdr:
            self.naxis = (int(hdr["ZNAXIS1"]), int(hdr["ZNAXIS2"]))
        lng0, lat0 = self.crval
        phi0, theta0 = 0.0, 0.0
        phip = float(hdr["LONPOLE"]) if "LONPOLE" in hdr else (0.0 if lat0 >= theta0 else 180.0)
        latpreq = float(hdr.get("LATPOLE", 90.0))
        # celset (cel.c:230-440), non-zenithal branch
        tol = 1.0e-10
        slat0, clat0 = math.sin(math.radians(lat0)), math.cos(math.radians(lat0))
        cthe0, sthe0 = math.cos(math.radians(theta0)), math.sin(math.radians(theta0))
        if phip == phi0:
            sphip, cphip = 0.0, 1.0
        else:
            sphip, cphip = math.sin(math.radians(phip - phi0)), math.cos(math.radians(phip - phi0))
        x, y = cthe0 * cphip, sthe0
        z = math.hypot(x, y)
        if z == 0.0:
            if slat0 != 0.0:
                raise InvalidTransformError("celset: invalid coordinate transformation parameters")
            latp = latpreq
        else:
            slz = slat0 / z
            if abs(slz) > 1.0:
                if abs(slz) - 1.0 < tol:
                    slz = math.copysign(1.0, slz)
                else:
                    raise InvalidTransformError("celset: invalid coordinate transformation parameters")
            uu = math.degrees(math.atan2(y, x))
            vv = math.degrees(math.acos(slz))
            latp1 = uu + vv
            if latp1 > 180.0:
                latp1 -= 360.0
            elif latp1 < -180.0:
                latp1 += 360.0
            latp2 = uu - vv
            if latp2 > 180.0:
                latp2 -= 360.0
            elif latp2 < -180.0:
                latp2 += 360.0
            if abs(latpreq - latp1) < abs(latpreq - latp2):
                latp = latp1 if abs(latp1) < 90.0 + tol else latp2
            else:
                latp = latp2 if abs(latp2) < 90.0 + tol else latp1
            if abs(latp) < 90.0 + tol:
                if latp > 90.0:
                    latp = 90.0
                elif latp < -90.0:
                    latp = -90.0
            else:
                raise InvalidTransformError("No valid solution for latp for these values of phip, phi0, and theta0")
        z = math.cos(math.radians(latp)) * clat0
        if abs(z) < tol:
            if abs(clat0) < tol:
                lngp = lng0
            elif latp > 0.0:
                lngp = lng0 + phip - phi0 - 180.0
            else:
                lngp = lng0 - phip + phi0
        else:
            xx = (sthe0 - math.sin(math.radians(latp)) * slat0) / z
            yy = sphip * cthe0 / clat0
            if xx == 0.0 and yy == 0.0:
                raise InvalidTransformError("celset: invalid coordinate transformation parameters")
            lngp = lng0 - math.degrees(math.atan2(yy, xx))
        if lng0 >= 0.0:
            if lngp < 0.0:
                lngp += 360.0
            elif lngp > 360.0:
                lngp -= 360.0
        else:
            if lngp > 0.0:
                lngp -= 360.0
            elif lngp < -360.0:
                lngp += 360.0
        self.euler = (lngp, 90.0 - latp, phip)  # celestial longitude of the native pole, its co-latitude, LONPOLE
        self.latp = latp

    def _matrix(self):
        return np.array([[self.cdelt[0] * self.pc[0, 0], self.cdelt[0] * self.pc[0, 1]],
                         [self.cdelt[1] * self.pc[1, 0], self.cdelt[1] * self.pc[1, 1]]])

    def pixel_to_world(self, px, py):
        """0-based pixel -> (lon, lat) degrees; longitude normalised as wcslib's sphx2s does (sign of the pole's
        celestial longitude decides [0, 360) or (-360, 0])."""
        px = np.asarray(px, dtype=np.float64)
        py = np.asarray(py, dtype=np.float64)
        q1 = px + 1.0 - self.crpix[0]
        q2 = py + 1.0 - self.crpix[1]
        m = self._matrix()
        phi = m[0, 0] * q1 + m[0, 1] * q2       # carx2s: phi = x, theta = y (degrees)
        theta = m[1, 0] * q1 + m[1, 1] * q2
        lngp, colat, phip = self.euler
        dphi = np.radians(phi - phip)
        th = np.radians(theta)
        cl, sl = math.cos(math.radians(colat)), math.sin(math.radians(colat))  # = sin(latp), cos(latp)
        sth, cth = np.sin(th), np.cos(th)
        x = sth * sl - cth * cl * np.cos(dphi)
        y = -cth * np.sin(dphi)
        z = sth * cl + cth * sl * np.cos(dphi)
        lng = lngp + np.degrees(np.arctan2(y, x))
        lat_a = np.degrees(np.arcsin(np.clip(z, -1.0, 1.0)))
        lat_b = np.copysign(np.degrees(np.arccos(np.clip(np.hypot(x, y), 0.0, 1.0))), z)
        lat = np.where(np.abs(z) > 0.99, lat_b, lat_a)
        if lngp >= 0.0:
            lng = np.where(lng < 0.0, lng + 360.0, lng)
        else:
            lng = np.where(lng > 0.0, lng - 360.0, lng)
        lng = np.where(lng > 360.0, lng - 360.0, lng)
        lng = np.where(lng < -360.0, lng + 360.0, lng)
        return lng, lat

    def world_to_pixel(self, lon, lat):
        """(lon, lat) degrees -> 0-based pixel (x, y)."""
        lon = np.asarray(lon, dtype=np.float64)
        lat = np.asarray(lat, dtype=np.float64)
        lngp, colat, phip = self.euler
        dl = np.radians(lon - lngp)
        la = np.radians(lat)
        cl, sl = math.cos(math.radians(colat)), math.sin(math.radians(colat))
        sla, cla = np.sin(la), np.cos(la)
        x = sla * sl - cla * cl * np.cos(dl)
        y = -cla * np.sin(dl)
        z = sla * cl + cla * sl * np.cos(dl)
        dphi = np.degrees(np.arctan2(y, x))
        phi = phip + dphi
        phi = np.where(phi > 180.0, phi - 360.0, phi)
        phi = np.where(phi < -180.0, phi + 360.0, phi)
        th_a = np.degrees(np.arcsin(np.clip(z, -1.0, 1.0)))
        th_b = np.copysign(np.degrees(np.arccos(np.clip(np.hypot(x, y), 0.0, 1.0))), z)
        theta = np.where(np.abs(z) > 0.99, th_b, th_a)
        mi = np.linalg.inv(self._matrix())
        q1 = mi[0, 0] * phi + mi[0, 1] * theta   # cars2x: x = phi, y = theta
        q2 = mi[1, 0] * phi + mi[1, 1] * theta
        return q1 + self.crpix[0] - 1.0, q2 + self.crpix[1] - 1.0


def make_wcs(hdr):
    """TAN or CAR by CTYPE1 (the two projections the alignment paths meet)."""
    ct = str(hdr.get("CTYPE1", "HPLN-TAN")).strip().upper()
    return CarWCS(hdr) if ct.endswith("-CAR") else TanWCS(hdr)


def extract_EUI_coordinates(hdr, wrap=True):
    """Longitude/latitude (degrees) of every pixel of `hdr`.  Util.py:282-312 (non-sunpy branch: list of Quantities;
    ang2pipi applied for HPLN-TAN, raw wcslib output for CRLN-CAR)."""
    w = make_wcs(hdr)
    x, y = np.meshgrid(np.arange(w.naxis[0]), np.arange(w.naxis[1]))
    lon, lat = w.pixel_to_world(x, y)
    if wrap and not isinstance(w, CarWCS):
        return ang2pipi(lon), ang2pipi(lat)
    return lon, lat


def extract_coordinates_pixels(header_initial_to_project, header_target_projection, order=None):
    """Pixel coordinates, in `header_target_projection`, of every pixel of
    `header_initial_to_project`.  alignment.py:1038-1069 (non-sunpy branch).  `order`: the spline order the
    coordinates will be sampled with (decides which coordinates are sensitive to wcslib's rounding noise)."""
    w_to = make_wcs(header_target_projection)
    lon, lat = extract_EUI_coordinates(header_initial_to_project)
    x, y = w_to.world_to_pixel(lon, lat)
    if isinstance(w_to, (TanWCS, CarWCS)) and "NAXIS1" in header_target_projection:
        # coordinates ON the bounds rule are decided by wcslib's rounding noise (see WcslibTan)
        x, y = np.array(x, dtype=np.float64), np.array(y, dtype=np.float64)
        wcslib_refine_near_bounds(header_initial_to_project, header_target_projection, x, y)
        if order is not None and int(order) % 2 == 1:
            wcslib_refine_near_integers(header_initial_to_project, header_target_projection, x, y)
    return x, y


# --------------------------------------------------------------------------------------
# header logic
def check_and_create_pcij_matrix(hdr, force_crota_0=False):
    """alignment.py:580-611."""
    if "PC1_1" not in hdr:
        if "CROTA" in hdr:
            crot = hdr["CROTA"]
        elif "CROTA2" in hdr:
            crot = hdr["CROTA2"]
        else:
            if force_crota_0:
                crot = 0.0
                hdr["CROTA"] = 0.0
            else:
                raise ValueError("No, CROTA, CROTA2 or PCi_j matrix in your FITS file.")
        rho = np.deg2rad(crot)
        lam = hdr["CDELT2"] / hdr["CDELT1"]
        hdr["PC1_1"] = np.cos(rho)
        hdr["PC2_2"] = np.cos(rho)
        hdr["PC1_2"] = -lam * np.sin(rho)
        hdr["PC2_1"] = (1 / lam) * np.sin(rho)
    if hdr["PC1_1"] >= 1.0:
        hdr["PC1_1"] = 1.0
        hdr["PC2_2"] = 1.0
        hdr["PC1_2"] = 0.0
        hdr["PC2_1"] = 0.0
        hdr["CROTA"] = 0.0
    if "CROTA" not in hdr:
        s = -np.sign(hdr["PC1_2"]) + (hdr["PC1_2"] == 0)
        hdr["CROTA"] = s * np.rad2deg(np.arccos(hdr["PC1_1"]))


class SweepState:
    """The attributes of `Alignment` that the sweep reads (alignment.py:78-140, 799-842)."""

    def __init__(self, hdr_small, hdr_large, data_small, data_large, lag_crval1, lag_crval2, lag_cdelt1,
                 lag_cdelt2, lag_crota, lag_solar_r=None, unit_lag="arcsec", order=2,
                 cdelt_semantics="intended"):
        self.hdr_small = dict(hdr_small)
        self.hdr_large = dict(hdr_large)
        self.data_small = np.array(data_small, dtype=np.float64)
        self.data_large = np.array(data_large, dtype=np.float64)

        def arr(v):
            return np.array([0.0]) if v is None else np.atleast_1d(np.asarray(v, dtype=np.float64))

        self.lag_crval1 = arr(lag_crval1)
        self.lag_crval2 = arr(lag_crval2)
        self.lag_cdelt1 = arr(lag_cdelt1)
        self.lag_cdelt2 = arr(lag_cdelt2)
        self.lag_crota = arr(lag_crota)
        self.lag_solar_r = None if lag_solar_r is None else np.atleast_1d(np.asarray(lag_solar_r, dtype=np.float64))
        self.unit_lag = unit_lag
        self.order = order
        # "reference": CDELT1 lag is a no-op, CDELT2 lag raises (alignment.py:420-440, quirk Q2)
        # "intended": CDELTi += d then PC rebuilt (Util.py:199-215)
        self.cdelt_semantics = cdelt_semantics
        self.lonlims = None
        self.latlims = None
        self.shape = None


def set_initial_header_values(st: SweepState, use_ang2pipi=True):
    """alignment.py:799-842."""
    h = st.hdr_small
    st.crval1_ref = h["CRVAL1"]
    st.crval2_ref = h["CRVAL2"]
    if "CROTA" in h:
        st.crota_ref = h["CROTA"]
    elif "CROTA2" in h:
        st.crota_ref = h["CROTA2"]
    else:
        s = -np.sign(h["PC1_2"]) + (h["PC1_2"] == 0)
        st.crota_ref = np.rad2deg(np.arccos(h["PC1_1"])) * s
        h["CROTA"] = np.rad2deg(np.arccos(h["PC1_1"]))
    st.cdelt1_ref = h["CDELT1"]
    st.cdelt2_ref = h["CDELT2"]
    st.unit1 = h["CUNIT1"]
    st.unit2 = h["CUNIT2"]
    if st.unit_lag in st.unit1:
        pass
    else:
        f1 = unit_to_deg(st.unit_lag) / unit_to_deg(st.unit1)
        f2 = unit_to_deg(st.unit_lag) / unit_to_deg(st.unit2)
        if use_ang2pipi:
            st.lag_crval1 = ang2pipi_unit(st.lag_crval1, st.unit_lag) * f1
            st.lag_crval2 = ang2pipi_unit(st.lag_crval2, st.unit_lag) * f2
            st.lag_cdelt1 = ang2pipi_unit(st.lag_cdelt1, st.unit_lag) * f1
            st.lag_cdelt2 = ang2pipi_unit(st.lag_cdelt2, st.unit_lag) * f2
        else:
            st.lag_crval1 = st.lag_crval1 * f1
            st.lag_crval2 = st.lag_crval2 * f2
            st.lag_cdelt1 = st.lag_cdelt1 * f1
            st.lag_cdelt2 = st.lag_cdelt2 * f2
        st.unit_lag = st.unit1
    if st.unit1 != st.unit2:
        raise ValueError("CUNIT1 and CUNIT2 must be equal")
    if st.lag_solar_r is None:
        st.lag_solar_r = np.array([1.004])


def shift_header(st: SweepState, hdr, d_crval1, d_crval2, d_cdelt1, d_cdelt2, d_crota):
    """alignment.py:401-468, including quirk Q3 (PC rebuilt only when one of d_crota,
    d_cdelt1, d_cdelt2 is != 0.0).  CDELT lags: `cdelt_semantics`."""
    if st.unit_lag != hdr["CUNIT1"] or st.unit_lag != hdr["CUNIT2"]:
        raise ValueError("lag.unit and cUNIT are not the same")
    hdr["CRVAL1"] = st.crval1_ref + d_crval1
    hdr["CRVAL2"] = st.crval2_ref + d_crval2
    change_pcij = False
    if d_cdelt1 != 0.0:
        change_pcij = True
        if st.cdelt_semantics == "intended":
            hdr["CDELT1"] = st.cdelt1_ref + d_cdelt1
        # "reference": computed but never written (alignment.py:423-430)
    if d_cdelt2 != 0.0:
        change_pcij = True
        if st.cdelt_semantics == "intended":
            hdr["CDELT2"] = st.cdelt2_ref + d_cdelt2
        else:
            raise AttributeError("'numpy.float64' object has no attribute 'to'")  # alignment.py:440
    if d_crota != 0.0:
        change_pcij = True
        if "CROTA" in hdr:
            hdr["CROTA"] = st.crota_ref + d_crota
        elif "CROTA2" in hdr:
            hdr["CROTA2"] = st.crota_ref + d_crota
        crot = st.crota_ref + d_crota
    else:
        crot = st.crota_ref
    if change_pcij:
        rho = np.deg2rad(crot)
        lam = hdr["CDELT2"] / hdr["CDELT1"]
        hdr["PC1_1"] = np.cos(rho)
        hdr["PC2_2"] = np.cos(rho)
        hdr["PC1_2"] = -lam * np.sin(rho)
        hdr["PC2_1"] = (1 / lam) * np.sin(rho)


def set_threshold_minmax_to_nan(data_small, vmin=None, vmax=None):
    """alignment.py:876-887 (in place)."""
    c1 = np.ones(data_small.shape, dtype=bool)
    c2 = np.ones(data_small.shape, dtype=bool)
    with np.errstate(invalid="ignore"):
        if vmin is not None:
            c1[np.abs(data_small) < vmin] = False
        if vmax is not None:
            c2[np.abs(data_small) > vmax] = False
    data_small[~(c1 & c2)] = np.nan


def set_remove_fov_limits_to_nan(st: "SweepState", lonlims_deg, latlims_deg):
    """alignment.py:863-874 (limits in degrees)."""
    lon, lat = extract_EUI_coordinates(st.hdr_small)
    inside = np.logical_and(np.logical_and(lon >= lonlims_deg[0], lon <= lonlims_deg[1]),
                            np.logical_and(lat >= latlims_deg[0], lat <= latlims_deg[1]))
    st.data_small[inside] = np.nan


def build_regular_grid(longitude, latitude, lonlims=None, latlims=None):
    """PlotFits.build_regular_grid, utils/Util.py:873-906, everything in degrees."""
    x = np.abs(longitude[0, 1] - longitude[0, 0])
    y = np.abs(latitude[0, 1] - latitude[0, 0])
    dlon = np.sqrt(x ** 2 + y ** 2)
    x = np.abs(longitude[1, 0] - longitude[0, 0])
    y = np.abs(latitude[1, 0] - latitude[0, 0])
    dlat = np.sqrt(x ** 2 + y ** 2)
    longitude1D = np.arange(np.min(longitude), np.max(longitude), dlon)
    latitude1D = np.arange(np.min(latitude), np.max(latitude), dlat)
    if (lonlims is not None) or (latlims is not None):
        longitude1D = longitude1D[(longitude1D > lonlims[0]) & (longitude1D < lonlims[1])]
        latitude1D = latitude1D[(latitude1D > latlims[0]) & (latitude1D < latlims[1])]
    long, latg = np.meshgrid(longitude1D, latitude1D)
    return long, latg, dlon, dlat


def select_fov_in_small_data(st: "SweepState", lonlims_deg, latlims_deg):
    """alignment.py:1082-1127 (limits in degrees), literal: CRPIX1/NAXIS1 take the ROW count of the regular grid."""
    lon, lat = extract_EUI_coordinates(st.hdr_small)
    long, latg, dlon, dlat = build_regular_grid(lon, lat, lonlims_deg, latlims_deg)
    mid = [long.shape[0] // 2, long.shape[1] // 2]
    hg = dict(st.hdr_small)
    u1, u2 = unit_to_deg(hg["CUNIT1"]), unit_to_deg(hg["CUNIT2"])
    hg["CRVAL1"] = long[mid[0], mid[1]] / u1
    hg["CRVAL2"] = latg[mid[0], mid[1]] / u2
    hg["CRPIX1"] = mid[0] + 1
    hg["CRPIX2"] = mid[1] + 1
    hg["CDELT1"] = dlon / u1
    hg["CDELT2"] = dlat / u2
    hg["PC1_1"], hg["PC2_2"], hg["PC1_2"], hg["PC2_1"] = 1.0, 1.0, 0.0, 0.0
    hg["CROTA"] = 0.0
    hg["CROTA2"] = 0.0
    hg["NAXIS1"] = long.shape[0]
    hg["NAXIS2"] = long.shape[1]
    xg, yg = extract_coordinates_pixels(hg, st.hdr_small)
    out = np.zeros_like(xg)
    interpol2d(st.data_small, x=xg, y=yg, order=st.order, fill=np.nan, dst=out)
    st.data_small = out
    st.hdr_small = hg


# --------------------------------------------------------------------------------------
# Carrington: rectify.py:282-423 (transforms), :842-888 (Rectifier), alignment.py:889-901
def carrington_grid(shape, lonlims, latlims, dtype=np.float32):
    """rectify.py:875-878: meshgrid(linspace(lon, shape[0], float32), linspace(lat, shape[1], float32));
    arrays have shape [shape[1], shape[0]]."""
    return np.meshgrid(np.linspace(lonlims[0], lonlims[1], shape[0], dtype=dtype),
                       np.linspace(latlims[0], latlims[1], shape[1], dtype=dtype))


def carrington_params(hdr, radius_correction):
    """rectify.py:387-415: header -> SphericalTransform arguments."""
    roll = hdr["CROTA"] if "CROTA" in hdr else hdr["CROTA2"]
    cos = np.cos(np.radians(roll))
    sin = np.sin(np.radians(roll))
    dx = cos * hdr["CRVAL1"] + sin * hdr["CRVAL2"]
    dy = -sin * hdr["CRVAL1"] + cos * hdr["CRVAL2"]
    return dict(x0=(hdr["CRPIX1"] - 1) - dx / hdr["CDELT1"],
                y0=(hdr["CRPIX2"] - 1) - dy / hdr["CDELT2"],
                dist=hdr["DSUN_OBS"] / (radius_correction * R_SUN_M),
                lon0=np.radians(hdr["CRLN_OBS"]), lat0=np.radians(hdr["CRLT_OBS"]),
                roll=np.radians(roll), cdelt1=hdr["CDELT1"], cdelt2=hdr["CDELT2"])


def carrington_coords(hdr, radius_correction, shape, lonlims, latlims):
    """(nx, ny) pixel coordinates in the image of `hdr` of every Carrington grid point.
    rectify.py:304-311 (differential rotation, dead: rate_wave=None => dx == 0 but promotes the
    longitude array to float64, quirk Q5) then rectify.py:340-363.  dtype flow is that of NumPy 2
    (NEP 50): grid fp32, radians/sin/cos of latitude in fp32, everything else fp64."""
    x, y = carrington_grid(shape, lonlims, latlims)
    # DifferentialRotationTransform.forward with coeffs (14.18, 0, 0)
    siny2 = np.sin(np.radians(y)) ** 2
    dx = np.float64(0.0) * (14.18 + siny2 * (0 + 0 * siny2) - 14.18)
    x = x - dx
    p = carrington_params(hdr, radius_correction)
    # SphericalTransform.forward
    lon = np.radians(x) - p["lon0"]
    lat = np.radians(y)
    xs = np.cos(lat) * np.sin(lon)
    ys = np.sin(lat)
    zs = np.cos(lat) * np.cos(lon)
    zz = zs * np.cos(p["lat0"]) + ys * np.sin(p["lat0"])
    yy = ys * np.cos(p["lat0"]) - zs * np.sin(p["lat0"])
    gd = zz >= 0
    yr = yy[gd] * np.cos(p["roll"]) - xs[gd] * np.sin(p["roll"])
    xr = xs[gd] * np.cos(p["roll"]) + yy[gd] * np.sin(p["roll"])
    z = p["dist"] - zz[gd]
    nx = np.full_like(lon, np.nan)
    ny = np.full_like(lon, np.nan)
    nx[gd] = p["x0"] + np.degrees(np.arctan(xr / z)) * 3600 / p["cdelt1"]
    ny[gd] = p["y0"] + np.degrees(np.arctan(yr / z)) * 3600 / p["cdelt2"]
    return nx, ny


def carrington_transform_fa(data, hdr, d_solar_r, shape, lonlims, latlims, order=2):
    """alignment.py:889-901: Rectifier(CarringtonTransform(hdr))(data, ..., fill=-32762), -32762 -> NaN."""
    nx, ny = carrington_coords(hdr, d_solar_r, shape, lonlims, latlims)
    image = interpol2d(data, nx, ny, fill=-32762, order=order)
    return np.where(image == -32762, np.nan, image)


# --------------------------------------------------------------------------------------
# helioprojective: alignment.py:987-1029
def create_submap_of_large_data(st: SweepState):
    """alignment.py:987-1016: large image resampled on the small header's own pixel grid, fp32;
    hdr_large := copy of hdr_small."""
    hdr_cut = dict(st.hdr_small)
    x_cut, y_cut = extract_coordinates_pixels(hdr_cut, st.hdr_large, st.order)
    image_large_cut = np.zeros_like(x_cut, dtype="float32")
    interpol2d(st.data_large.copy(), x=x_cut, y=y_cut, dst=image_large_cut, order=st.order, fill=np.nan)
    st.hdr_large = dict(hdr_cut)
    return np.array(image_large_cut)


def interpolate_on_large_data_grid(st: SweepState, data, hdr):
    """alignment.py:1018-1029: small image resampled on hdr_large's pixel grid, fp32."""
    x_large, y_large = extract_coordinates_pixels(st.hdr_large, hdr, st.order)
    image_small_shft = np.zeros_like(x_large, dtype="float32")
    interpol2d(data.copy(), x=x_large, y=y_large, order=st.order, fill=np.nan, dst=image_small_shft)
    return image_small_shft


# --------------------------------------------------------------------------------------
# one lag-point: alignment.py:509-578
def masked_pearson(data_large, data_small_interp):
    """alignment.py:525-531."""
    a = data_large.ravel()
    b = data_small_interp.ravel()
    is_nan = np.logical_or(~np.isfinite(a), ~np.isfinite(b))
    A = np.array(a[~is_nan], dtype="float")
    B = np.array(b[~is_nan], dtype="float")
    return c_correlate(A, B, lags=[0])[0]


def residus(data_large, data_small_interp):
    """alignment.py:544-547 (no NaN mask: quirk Q8)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        norm = np.sqrt(data_large.ravel())
        diff = (data_large.ravel() - data_small_interp.ravel()) / norm
        return np.std(diff)


def step(st: SweepState, frame, data_small, data_large, d_crval1, d_crval2, d_cdelt1, d_cdelt2, d_crota,
         d_solar_r, method="correlation"):
    """alignment.py:509-549 / 551-578."""
    hdr = dict(st.hdr_small)
    shift_header(st, hdr, d_crval1, d_crval2, d_cdelt1, d_cdelt2, d_crota)
    if frame == "carrington":
        interp = carrington_transform_fa(data_small, hdr, d_solar_r, st.shape, st.lonlims, st.latlims, st.order)
    else:
        try:
            interp = interpolate_on_large_data_grid(st, data_small, hdr)
        except InvalidTransformError:
            # CAR inputs: a CRVAL2 lag can leave no valid native pole for an explicit LONPOLE; astropy raises, the
            # reference's worker dies and the slot keeps its initial value -- reported as NaN here (quirk Q9)
            return np.nan
    if method == "correlation":
        return masked_pearson(data_large, interp)
    elif method == "residus":
        return residus(data_large, interp)
    raise NotImplementedError


def prepare_reference(st: SweepState, frame, d_solar_r, parallelism=True):
    """alignment.py:646-651 (parallel) / 762-767 (serial; quirk Q1: helioprojective keeps the
    FULL large grid because the condition tests 'initial_helioprojective').  frame 'initial_carrington'
    (align_using_initial_carrington, alignment.py:344-399) is named by BOTH conditions: always the sub-map."""
    if frame == "carrington":
        return carrington_transform_fa(st.data_large, st.hdr_large, d_solar_r, st.shape, st.lonlims,
                                       st.latlims, st.order)
    if parallelism or frame == "initial_carrington":
        return create_submap_of_large_data(st)
    return st.data_large


def lag_table(st: SweepState):
    """C-order ravel of meshgrid(crval1, crval2, cdelt1, cdelt2, crota, indexing='ij').
    alignment.py:667-674."""
    g = np.meshgrid(st.lag_crval1, st.lag_crval2, st.lag_cdelt1, st.lag_cdelt2, st.lag_crota, indexing="ij")
    return np.stack([a.ravel() for a in g], axis=1), g[0].shape


def _worker(args):
    (shm_small, shp_small, shm_large, shp_large, dt_large, st, frame, lags, d_solar_r, method) = args
    s1 = shared_memory.SharedMemory(name=shm_small)
    s2 = shared_memory.SharedMemory(name=shm_large)
    try:
        data_small = np.ndarray(shp_small, dtype=np.float64, buffer=s1.buf)
        data_large = np.ndarray(shp_large, dtype=dt_large, buffer=s2.buf)
        out = np.zeros(len(lags))
        for i, lg in enumerate(lags):
            out[i] = step(st, frame, data_small, data_large, lg[0], lg[1], lg[2], lg[3], lg[4], d_solar_r, method)
        return out
    finally:
        s1.close()
        s2.close()


def find_best_header_parameters(st: SweepState, frame, method="correlation", parallelism=True, counts=None,
                                lag_subset=None, prepared_reference=None, use_ang2pipi=True, reference_quirks=False):
    """alignment.py:613-797.  Returns the 6-D corr array [crval1, crval2, cdelt1, cdelt2, crota, solar_r].
    `frame`: 'carrington', 'helioprojective' or 'initial_carrington' (CAR maps on both sides; sub-map always).
    `use_ang2pipi`: False for align_using_initial_carrington (alignment.py:388).
    `reference_quirks`: reproduce quirk Q10 -- the serial branch overwrites `self.data_large` with the prepared
    reference (alignment.py:763-764), so a second `lag_solar_r` value re-projects the already re-projected image.
    `counts` > 1 fans the raveled lag list out over processes in `np.array_split` chunks with the images in
    POSIX shared memory (alignment.py:667-744); `counts` in (None, 1) runs in-process.
    `lag_subset` (indices into the raveled lag list) restricts the computation (bench sampling);
    other entries are NaN.  `prepared_reference`: the reference image already on the target grid (skips the
    once-only preparation, alignment.py:646-651, so that it can be kept out of a timed region)."""
    set_initial_header_values(st, use_ang2pipi)
    table, shp = lag_table(st)
    nsr = len(st.lag_solar_r)
    corr = np.full((table.shape[0], nsr), np.nan)
    for kk, d_solar_r in enumerate(st.lag_solar_r):
        if prepared_reference is not None:
            data_large = prepared_reference
            if frame != "carrington" and parallelism:
                st.hdr_large = dict(st.hdr_small)  # alignment.py:1000
        else:
            data_large = prepare_reference(st, frame, d_solar_r, parallelism)
            if reference_quirks and not parallelism and frame == "carrington":
                st.data_large = data_large
        if np.isnan(st.data_small).all():
            raise ValueError("minimum or maximum value have set all small FOV to nan")
        idx = np.arange(table.shape[0]) if lag_subset is None else np.asarray(lag_subset)
        if counts is None or counts <= 1:
            for i in idx:
                lg = table[i]
                corr[i, kk] = step(st, frame, st.data_small, data_large, lg[0], lg[1], lg[2], lg[3], lg[4],
                                   d_solar_r, method)
        else:
            data_large = np.ascontiguousarray(data_large)
            s1 = shared_memory.SharedMemory(create=True, size=st.data_small.nbytes)
            s2 = shared_memory.SharedMemory(create=True, size=data_large.nbytes)
            try:
                np.ndarray(st.data_small.shape, dtype=np.float64, buffer=s1.buf)[...] = st.data_small
                np.ndarray(data_large.shape, dtype=data_large.dtype, buffer=s2.buf)[...] = data_large
                chunks = [c for c in np.array_split(idx, counts) if len(c)]
                st_light = copy.copy(st)
                st_light.data_small = None
                st_light.data_large = None
                jobs = [(s1.name, st.data_small.shape, s2.name, data_large.shape, data_large.dtype, st_light, frame,
                         table[c], d_solar_r, method) for c in chunks]
                with mp.get_context("fork").Pool(len(chunks)) as pool:
                    outs = pool.map(_worker, jobs)
                for c, o in zip(chunks, outs):
                    corr[c, kk] = o
            finally:
                s1.close()
                s1.unlink()
                s2.close()
                s2.unlink()
    return corr.reshape(shp + (nsr,))


# --------------------------------------------------------------------------------------
# hdrshift/AlignmentResults.py:12-21, 218-341
def twoD_Gaussian(xy, amplitude, xo, yo, sigma_x, sigma_y, offset):
    x, y = xy
    g = offset + amplitude * np.exp(-((((x - float(xo)) ** 2) / (2 * sigma_x ** 2))
                                      + (((y - float(yo)) ** 2) / (2 * sigma_y ** 2))))
    return g.ravel()


def compute_shift(corr, lag_crval1_arcsec, lag_crval2_arcsec):
    """AlignmentResults.py:218-341 restated: returns (max_index, shift_pixels(x,y), shift_arcsec(x,y))."""
    from scipy.optimize import curve_fit
    max_index = np.unravel_index(np.nanargmax(corr), corr.shape)
    corr2d = corr[:, :, max_index[2], max_index[3], max_index[4]]
    px = [max_index[0]]
    py = [max_index[1]]
    lenx, leny = corr2d.shape[0], corr2d.shape[1]
    for ii in (-2, -1, 0, 1, 2):
        for jj in (-2, -1, 0, 1, 2):
            x = max_index[0] + ii
            y = max_index[1] + jj
            if (x != -1) and (x < lenx) and (y != -1) and (y < leny):
                px.append(x)
                py.append(y)
    if len(px) < 4:
        return max_index, (max_index[0], max_index[1]), (lag_crval1_arcsec[max_index[0]], lag_crval2_arcsec[max_index[1]])
    A = (np.float64(px), np.float64(py))
    B = np.float64(corr2d[px, py].ravel())
    p0 = (np.float64(corr2d[max_index[0], max_index[1]][0]), np.float64(max_index[0]), np.float64(max_index[1]),
          1.0, 1.0, 0.9)
    bounds = ([0.0, max_index[0] - 5.0, max_index[1] - 5.0, 0.0, 0.0, -10.0],
              [10.0, max_index[0] + 5.0, max_index[1] + 5.0, 1000.0, 1000.0, 10.0])
    try:
        popt, _ = curve_fit(f=twoD_Gaussian, xdata=A, ydata=B, p0=p0, bounds=bounds)
    except ValueError:
        return max_index, (max_index[0], max_index[1]), (lag_crval1_arcsec[max_index[0]], lag_crval2_arcsec[max_index[1]])
    sx = np.interp(popt[1], np.arange(len(lag_crval1_arcsec)), lag_crval1_arcsec)
    sy = np.interp(popt[2], np.arange(len(lag_crval2_arcsec)), lag_crval2_arcsec)
    return max_index, (popt[1], popt[2]), (sx, sy)
