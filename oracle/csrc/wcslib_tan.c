/*
 * TEST INFRASTRUCTURE (oracle): the scalar, libm-based restatement of wcslib's TAN pixel -> sky -> pixel chain that
 * oracle/coreg_oracle.py holds as the Python class `WcslibTan`, in plain C, so that the oracle can re-evaluate EVERY
 * pixel of a 2048^2 grid (odd spline orders at noise-decided lag-points, coreg_oracle.wcslib_refine_near_integers)
 * instead of stopping at the 300 000 pixels a pure-Python loop manages.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product never does.
 *
 * Restated from wcslib 7.x (third-party, bundled with astropy; absent from the reference tree): lin.c linp2x / linx2p /
 * matinv, prj.c tanx2s / tans2x, sph.c sphx2s / sphs2x, wcstrig.c, with the macro expansions that fix the order of the
 * roundings (angle*D2R is (angle*PI)/180, atan2(y,x)*R2D is (atan2(y,x)*180)/PI).  The reference reaches this chain
 * through astropy at hdrshift/alignment.py:1038-1069 and utils/Util.py:282-312 (ang2pipi between the two halves,
 * utils/Util.py:76-80).  Operation order follows coreg_oracle.WcslibTan line by line; the two are checked against each
 * other and against astropy 4.3.1 / wcslib 7.6 bit for bit (tests/test_oracle_golden.py, tests/golden/border_golden.npz).
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off: no fused multiply-add, no value-changing optimisation).
 */
#define _GNU_SOURCE /* sincos() */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define WT_PI 3.141592653589793238462643
static const double R2D = 180.0 / WT_PI;

static double cosd(double a) {
    if (fmod(a, 90.0) == 0.0) {
        static const double t[4] = {1.0, 0.0, -1.0, 0.0};
        long i = labs((long)floor(a / 90.0 + 0.5)) % 4;
        return t[i];
    }
    return cos(a * WT_PI / 180.0);
}
static double sind(double a) {
    if (fmod(a, 90.0) == 0.0) {
        static const double t[4] = {1.0, 0.0, -1.0, 0.0};
        long i = labs((long)floor(a / 90.0 - 0.5)) % 4;
        return t[i];
    }
    return sin(a * WT_PI / 180.0);
}
static void sincosd(double a, double* s, double* c) {
    if (fmod(a, 90.0) == 0.0) {
        long i = labs((long)floor(a / 90.0 + 0.5)) % 4;
        if (i == 0) { *s = 0.0; *c = 1.0; }
        else if (i == 1) { *s = a > 0.0 ? 1.0 : -1.0; *c = 0.0; }
        else if (i == 2) { *s = 0.0; *c = -1.0; }
        else { *s = a > 0.0 ? -1.0 : 1.0; *c = 0.0; }
        return;
    }
    sincos(a * WT_PI / 180.0, s, c);
}
static double atan2d(double y, double x) {
    if (y == 0.0) {
        if (x >= 0.0) return 0.0;
        if (x < 0.0) return 180.0;
    } else if (x == 0.0) {
        if (y > 0.0) return 90.0;
        if (y < 0.0) return -90.0;
    }
    return atan2(y, x) * 180.0 / WT_PI;
}
static double asind(double v) {
    if (v <= -1.0) {
        if (v + 1.0 > -1e-12) return -90.0;
    } else if (v == 0.0) {
        return 0.0;
    } else if (v >= 1.0) {
        if (v - 1.0 < 1e-12) return 90.0;
    }
    return asin(v) * 180.0 / WT_PI;
}
static double acosd(double v) {
    if (v >= 1.0) {
        if (v - 1.0 < 1e-10) return 0.0;
    } else if (v == 0.0) {
        return 90.0;
    } else if (v <= -1.0) {
        if (v + 1.0 > -1e-10) return 180.0;
    }
    return acos(v) * 180.0 / WT_PI;
}

/* lin.c matinv() for n = 2: LU with scaled partial pivoting, then a column-by-column solve */
static void matinv2(const double m[2][2], double inv[2][2]) {
    const int n = 2;
    int mxl[2] = {0, 1}, lxm[2] = {0, 0};
    double rowmax[2] = {0.0, 0.0};
    double lu[2][2] = {{m[0][0], m[0][1]}, {m[1][0], m[1][1]}};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double d = fabs(m[i][j]);
            if (d > rowmax[i]) rowmax[i] = d;
        }
    for (int k = 0; k < n; ++k) {
        double colmax = fabs(lu[k][k]) / rowmax[k];
        int pivot = k;
        for (int i = k + 1; i < n; ++i) {
            const double d = fabs(lu[i][k]) / rowmax[i];
            if (d > colmax) { colmax = d; pivot = i; }
        }
        if (pivot > k) {
            for (int j = 0; j < n; ++j) { const double t = lu[pivot][j]; lu[pivot][j] = lu[k][j]; lu[k][j] = t; }
            { const double t = rowmax[pivot]; rowmax[pivot] = rowmax[k]; rowmax[k] = t; }
            { const int t = mxl[pivot]; mxl[pivot] = mxl[k]; mxl[k] = t; }
        }
        for (int i = k + 1; i < n; ++i)
            if (lu[i][k] != 0.0) {
                lu[i][k] /= lu[k][k];
                for (int j = k + 1; j < n; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
            }
    }
    for (int i = 0; i < n; ++i) lxm[mxl[i]] = i;
    inv[0][0] = inv[0][1] = inv[1][0] = inv[1][1] = 0.0;
    for (int k = 0; k < n; ++k) {
        inv[lxm[k]][k] = 1.0;
        for (int i = lxm[k] + 1; i < n; ++i)
            for (int j = lxm[k]; j < i; ++j) inv[i][k] -= lu[i][j] * inv[j][k];
        for (int i = n - 1; i >= 0; --i) {
            for (int j = i + 1; j < n; ++j) inv[i][k] -= lu[i][j] * inv[j][k];
            inv[i][k] /= lu[i][i];
        }
    }
}

typedef struct {
    double crpix[2], cdelt[2], piximg[2][2], imgpix[2][2];
    int unity;
    double e0, e1, e2, e3, e4; /* celset, zenithal: (lng0, 90 - lat0, phiP, cos(e1), sin(e1)) */
} wcstan;

/* par = crpix1, crpix2, cdelt1 [deg], cdelt2 [deg], crval1 [deg], crval2 [deg], pc1_1, pc1_2, pc2_1, pc2_2, lonpole */
static void wcstan_init(wcstan* w, const double* par) {
    w->crpix[0] = par[0]; w->crpix[1] = par[1];
    w->cdelt[0] = par[2]; w->cdelt[1] = par[3];
    w->unity = par[6] == 1.0 && par[9] == 1.0 && par[7] == 0.0 && par[8] == 0.0;
    w->piximg[0][0] = par[2] * par[6]; w->piximg[0][1] = par[2] * par[7];
    w->piximg[1][0] = par[3] * par[8]; w->piximg[1][1] = par[3] * par[9];
    if (!w->unity) matinv2(w->piximg, w->imgpix);
    w->e0 = par[4];
    w->e1 = 90.0 - par[5];
    w->e2 = par[10];
    sincosd(w->e1, &w->e4, &w->e3);
}

static void wcstan_p2s(const wcstan* w, double px0, double py0, double* lng_out, double* lat_out) {
    const double t0 = (px0 + 1.0) - w->crpix[0];
    const double t1 = (py0 + 1.0) - w->crpix[1];
    double x, y;
    if (w->unity) {
        x = w->cdelt[0] * t0;
        y = w->cdelt[1] * t1;
    } else {
        x = 0.0; y = 0.0;
        x += w->piximg[0][0] * t0;
        y += w->piximg[1][0] * t0;
        x += w->piximg[0][1] * t1;
        y += w->piximg[1][1] * t1;
    }
    const double xj = x + 0.0, yj = y + 0.0;
    const double r = sqrt(xj * xj + yj * yj);
    const double phi = r == 0.0 ? 0.0 : atan2d(xj, -yj);
    const double theta = atan2d(R2D, r);
    const double dphi = phi - w->e2;
    double sinthe, costhe, sinphi, cosphi;
    sincosd(theta, &sinthe, &costhe);
    const double costhe3 = costhe * w->e3, costhe4 = costhe * w->e4;
    const double sinthe3 = sinthe * w->e3, sinthe4 = sinthe * w->e4;
    sincosd(dphi, &sinphi, &cosphi);
    double xx = sinthe4 - costhe3 * cosphi;
    if (fabs(xx) < 1e-5) xx = -cosd(theta + w->e1) + costhe3 * (1.0 - cosphi);
    const double yy = -costhe * sinphi;
    double dlng;
    if (xx != 0.0 || yy != 0.0) dlng = atan2d(yy, xx);
    else dlng = dphi + 180.0;
    double lng = w->e0 + dlng;
    if (w->e0 >= 0.0) {
        if (lng < 0.0) lng += 360.0;
    } else {
        if (lng > 0.0) lng -= 360.0;
    }
    if (lng > 360.0) lng -= 360.0;
    else if (lng < -360.0) lng += 360.0;
    double lat;
    if (fmod(dphi, 180.0) == 0.0) {
        lat = theta + cosphi * w->e1;
        if (lat > 90.0) lat = 180.0 - lat;
        if (lat < -90.0) lat = -180.0 - lat;
    } else {
        const double z = sinthe3 + costhe4 * cosphi;
        if (fabs(z) > 0.99) lat = copysign(acosd(sqrt(xx * xx + yy * yy)), z);
        else lat = asind(z);
    }
    *lng_out = lng;
    *lat_out = lat;
}

static void wcstan_s2p(const wcstan* w, double lng, double lat, double* ox, double* oy) {
    const double dlng = lng - w->e0;
    double sinlat, coslat, sinlng, coslng;
    sincosd(lat, &sinlat, &coslat);
    const double coslat3 = coslat * w->e3, coslat4 = coslat * w->e4;
    const double sinlat3 = sinlat * w->e3, sinlat4 = sinlat * w->e4;
    sincosd(dlng, &sinlng, &coslng);
    double xx = sinlat4 - coslat3 * coslng;
    if (fabs(xx) < 1e-5) xx = -cosd(lat + w->e1) + coslat3 * (1.0 - coslng);
    const double yy = -coslat * sinlng;
    double dphi;
    if (xx != 0.0 || yy != 0.0) dphi = atan2d(yy, xx);
    else dphi = dlng - 180.0;
    double phi = fmod(w->e2 + dphi, 360.0);
    if (phi > 180.0) phi -= 360.0;
    else if (phi < -180.0) phi += 360.0;
    double theta;
    if (fmod(dlng, 180.0) == 0.0) {
        theta = lat + coslng * w->e1;
        if (theta > 90.0) theta = 180.0 - theta;
        if (theta < -90.0) theta = -180.0 - theta;
    } else {
        const double z = sinlat3 + coslat4 * coslng;
        if (fabs(z) > 0.99) theta = copysign(acosd(sqrt(xx * xx + yy * yy)), z);
        else theta = asind(z);
    }
    double sinphi, cosphi;
    sincosd(phi, &sinphi, &cosphi);
    const double s = sind(theta);
    if (s == 0.0 || s < 0.0) {
        *ox = *oy = NAN;
        return;
    }
    const double r = R2D * cosd(theta) / s;
    const double x = r * sinphi - 0.0;
    const double y = -r * cosphi - 0.0;
    double p0, p1;
    if (w->unity) {
        p0 = x / w->cdelt[0] + w->crpix[0];
        p1 = y / w->cdelt[1] + w->crpix[1];
    } else {
        p0 = 0.0;
        p0 += w->imgpix[0][0] * x;
        p0 += w->imgpix[0][1] * y;
        p0 += w->crpix[0];
        p1 = 0.0;
        p1 += w->imgpix[1][0] * x;
        p1 += w->imgpix[1][1] * y;
        p1 += w->crpix[1];
    }
    *ox = p0 - 1.0;
    *oy = p1 - 1.0;
}

/* utils/Util.py:76-80 on a float64: -((-a + 180) % 360 - 180) with NumPy's floored modulo (npy_mod) */
static double npy_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}
static double ang2pipi(double a) { return -(npy_mod(-a + 180.0, 360.0) - 180.0); }

/* pixels (0-based) of header `from` -> sky -> ang2pipi -> pixels (0-based) of header `to`; lng / lat optional */
int oracle_wcstan_pixel_to_pixel(const double* par_from, const double* par_to, int64_t n, const double* px,
                                 const double* py, double* ox, double* oy, double* lng_out, double* lat_out) {
    wcstan wf, wt;
    wcstan_init(&wf, par_from);
    wcstan_init(&wt, par_to);
    for (int64_t k = 0; k < n; ++k) {
        double lng, lat;
        wcstan_p2s(&wf, px[k], py[k], &lng, &lat);
        if (lng_out) lng_out[k] = lng;
        if (lat_out) lat_out[k] = lat;
        wcstan_s2p(&wt, ang2pipi(lng), ang2pipi(lat), &ox[k], &oy[k]);
    }
    return 0;
}
