// Host-side geometry of the alignment sweep: header shifting, TAN->TAN homographies, Carrington
// transform parameters and the float32 lon/lat grid tables.  Pure C++ (no HIP), shared by the C-ABI.
//
// Reference behaviour restated here (paths relative to the reference repo, euispice_coreg/...):
//   hdrshift/alignment.py:401-468   _shift_header
//   hdrshift/alignment.py:1038-1069 _extract_coordinates_pixels (pixel -> world -> pixel through astropy.wcs)
//   utils/rectify.py:377-423        CarringtonTransform (header -> SphericalTransform arguments)
//   utils/rectify.py:875-878        Rectifier grid (float32 linspace)
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/coreg_hip.h"

namespace coreg {

constexpr double kPi = 3.14159265358979323846;
constexpr double kDeg2Rad = kPi / 180.0;  // numpy: NPY_PI / 180.0
constexpr double kRad2Deg = 180.0 / kPi;  // numpy: 180.0 / NPY_PI
constexpr double kRSun = 695700000.0;     // astropy.constants.R_sun.value, utils/rectify.py:405

typedef long double ld;
struct Mat3 {
    ld m[3][3];
};
inline Mat3 mat_mul(const Mat3& a, const Mat3& b) {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            ld s = 0;
            for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline Mat3 mat_T(const Mat3& a) {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}

// ---- _shift_header, alignment.py:401-468 --------------------------------------------------------------
// Returns 0, or 1 when the reference's worker would die on this lag (COREG_CDELT_REFERENCE with d_cdelt2 != 0,
// alignment.py:440): the caller reports NaN for it.
inline int shift_header(const coreg_wcs2d& ref, double d_crval1, double d_crval2, double d_cdelt1, double d_cdelt2,
                        double d_crota, int cdelt_semantics, coreg_wcs2d* out) {
    *out = ref;
    out->crval1 = ref.crval1 + d_crval1;  // :404
    out->crval2 = ref.crval2 + d_crval2;  // :412
    bool change_pcij = false;
    if (d_cdelt1 != 0.0) {  // :421
        change_pcij = true;
        if (cdelt_semantics == COREG_CDELT_INTENDED) out->cdelt1 = ref.cdelt1 + d_cdelt1;
    }
    if (d_cdelt2 != 0.0) {  // :432
        change_pcij = true;
        if (cdelt_semantics == COREG_CDELT_INTENDED)
            out->cdelt2 = ref.cdelt2 + d_cdelt2;
        else
            return 1;
    }
    double crot = ref.crota;
    if (d_crota != 0.0) {  // :442
        change_pcij = true;
        out->crota = ref.crota + d_crota;
        crot = ref.crota + d_crota;
    }
    if (change_pcij) {  // :462-468
        const double rho = crot * kDeg2Rad;
        const double lam = out->cdelt2 / out->cdelt1;
        out->pc1_1 = std::cos(rho);
        out->pc2_2 = std::cos(rho);
        out->pc1_2 = -lam * std::sin(rho);
        out->pc2_1 = (1.0 / lam) * std::sin(rho);
    }
    return 0;
}

// ---- TAN WCS as matrices ------------------------------------------------------------------------------
// 0-based pixel p -> intermediate world (x, y) [radians]:  (x, y) = s * diag(cdelt_deg) * PC * (p + 1 - crpix)
inline Mat3 pix_to_iwc(const coreg_wcs2d& w) {
    const ld s = (ld)kPi / 180.0L;
    const ld c1 = (ld)w.cdelt1 * (ld)w.unit_to_deg * s, c2 = (ld)w.cdelt2 * (ld)w.unit_to_deg * s;
    const ld o1 = 1.0L - (ld)w.crpix1, o2 = 1.0L - (ld)w.crpix2;
    Mat3 a;
    a.m[0][0] = c1 * w.pc1_1; a.m[0][1] = c1 * w.pc1_2; a.m[0][2] = c1 * (w.pc1_1 * o1 + w.pc1_2 * o2);
    a.m[1][0] = c2 * w.pc2_1; a.m[1][1] = c2 * w.pc2_2; a.m[1][2] = c2 * (w.pc2_1 * o1 + w.pc2_2 * o2);
    a.m[2][0] = 0; a.m[2][1] = 0; a.m[2][2] = 1;
    return a;
}
inline Mat3 iwc_to_pix(const coreg_wcs2d& w) {
    const ld s = (ld)kPi / 180.0L;
    const ld c1 = (ld)w.cdelt1 * (ld)w.unit_to_deg * s, c2 = (ld)w.cdelt2 * (ld)w.unit_to_deg * s;
    const ld m00 = c1 * w.pc1_1, m01 = c1 * w.pc1_2, m10 = c2 * w.pc2_1, m11 = c2 * w.pc2_2;
    const ld det = m00 * m11 - m01 * m10;
    Mat3 a;
    a.m[0][0] = m11 / det; a.m[0][1] = -m01 / det; a.m[0][2] = (ld)w.crpix1 - 1.0L;
    a.m[1][0] = -m10 / det; a.m[1][1] = m00 / det; a.m[1][2] = (ld)w.crpix2 - 1.0L;
    a.m[2][0] = 0; a.m[2][1] = 0; a.m[2][2] = 1;
    return a;
}
// native direction n = (cos t cos p, cos t sin p, sin t) of a gnomonic point (x, y):  n ~ (-y, x, 1)
// native -> celestial rotation (FITS WCS paper II eq. 2, zenithal: pole = reference point):
//   R = Rz(alpha_p) * T(delta_p) * Rz(-phi_p),  T = [[-sin d, 0, cos d], [0, -1, 0], [cos d, 0, sin d]]
inline Mat3 native_to_celestial(const coreg_wcs2d& w) {
    const ld ap = (ld)w.crval1 * (ld)w.unit_to_deg * ((ld)kPi / 180.0L);
    const ld dp = (ld)w.crval2 * (ld)w.unit_to_deg * ((ld)kPi / 180.0L);
    const ld pp = (ld)w.lonpole * ((ld)kPi / 180.0L);
    Mat3 rz1 = {{{cosl(ap), -sinl(ap), 0}, {sinl(ap), cosl(ap), 0}, {0, 0, 1}}};
    Mat3 t = {{{-sinl(dp), 0, cosl(dp)}, {0, -1, 0}, {cosl(dp), 0, sinl(dp)}}};
    Mat3 rz2 = {{{cosl(pp), sinl(pp), 0}, {-sinl(pp), cosl(pp), 0}, {0, 0, 1}}};  // Rz(-phi_p)
    return mat_mul(rz1, mat_mul(t, rz2));
}

// Homography taking 0-based pixels of `from` to 0-based pixels of `to` through the common sky:
// what WCS(to).world_to_pixel(ang2pipi(WCS(from).pixel_to_world(p))) evaluates with per-pixel trig
// (alignment.py:1041-1065, Util.py:284-301).  H is scaled so that h[8] = 1.
inline void homography(const coreg_wcs2d& from, const coreg_wcs2d& to, double h[9]) {
    const Mat3 N = {{{0, -1, 0}, {1, 0, 0}, {0, 0, 1}}};
    const Mat3 Ninv = {{{0, 1, 0}, {-1, 0, 0}, {0, 0, 1}}};
    Mat3 m = mat_mul(N, pix_to_iwc(from));
    m = mat_mul(native_to_celestial(from), m);
    m = mat_mul(mat_T(native_to_celestial(to)), m);
    m = mat_mul(Ninv, m);
    m = mat_mul(iwc_to_pix(to), m);
    const ld s = m.m[2][2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) h[3 * i + j] = (double)(m.m[i][j] / s);
}
// The same homography for a whole lag sweep, factored so that the per-lag cost is two 3x3 products and no trig:
//   H(lag) = [iwc_to_pix(to) * Ninv] * [Rz(phi_p) * T(delta)] * [Rz(-alpha) * R(from) * N * pix_to_iwc(from)]
//             B: per (cdelt, crota) combo    P: depends on CRVAL2 only     Q: depends on CRVAL1 only
// (native_to_celestial(to)^T = Rz(phi_p) * T(delta_p) * Rz(-alpha_p), T symmetric).  alpha = CRVAL1 + lag1[i1],
// delta = CRVAL2 + lag2[i2] in the shifted header (alignment.py:404, :412).
struct Mat3d {
    double m[3][3];
};
inline Mat3d to_double(const Mat3& a) {
    Mat3d r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = (double)a.m[i][j];
    return r;
}
inline Mat3d mul3(const Mat3d& a, const Mat3d& b) {
    Mat3d r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
struct HomographyFamily {
    std::vector<Mat3d> Q;  // [n1]
    std::vector<Mat3d> P;  // [n2]
    // from: header whose pixel grid is mapped (target grid); ref: unshifted header being lagged
    void init(const coreg_wcs2d& from, const coreg_wcs2d& ref, const double* lag1, int n1, const double* lag2,
              int n2) {
        const Mat3 N = {{{0, -1, 0}, {1, 0, 0}, {0, 0, 1}}};
        const Mat3 m1 = mat_mul(native_to_celestial(from), mat_mul(N, pix_to_iwc(from)));
        const ld d2r = (ld)kPi / 180.0L;
        Q.resize(n1);
        for (int i = 0; i < n1; ++i) {
            const ld ap = (ld)(ref.crval1 + lag1[i]) * (ld)ref.unit_to_deg * d2r;  // float64 sum as alignment.py:404
            const Mat3 rz = {{{cosl(ap), sinl(ap), 0}, {-sinl(ap), cosl(ap), 0}, {0, 0, 1}}};  // Rz(-alpha)
            Q[i] = to_double(mat_mul(rz, m1));
        }
        const ld pp = (ld)ref.lonpole * d2r;
        const Mat3 rzp = {{{cosl(pp), -sinl(pp), 0}, {sinl(pp), cosl(pp), 0}, {0, 0, 1}}};  // Rz(phi_p)
        P.resize(n2);
        for (int j = 0; j < n2; ++j) {
            const ld dp = (ld)(ref.crval2 + lag2[j]) * (ld)ref.unit_to_deg * d2r;  // alignment.py:412
            const Mat3 t = {{{-sinl(dp), 0, cosl(dp)}, {0, -1, 0}, {cosl(dp), 0, sinl(dp)}}};
            P[j] = to_double(mat_mul(rzp, t));
        }
    }
    // per (cdelt, crota) combination: B = iwc_to_pix(shifted header) * Ninv
    static Mat3d combo(const coreg_wcs2d& shifted) {
        const Mat3 Ninv = {{{0, 1, 0}, {-1, 0, 0}, {0, 0, 1}}};
        return to_double(mat_mul(iwc_to_pix(shifted), Ninv));
    }
    void get(const Mat3d& B, int i1, int i2, double h[9]) const {
        const Mat3d m = mul3(B, mul3(P[i2], Q[i1]));
        const double s = 1.0 / m.m[2][2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) h[3 * i + j] = m.m[i][j] * s;
        h[8] = 1.0;  // exactly (the series form of the device map relies on it)
    }
};

inline void apply_h(const double h[9], double x, double y, double* ox, double* oy) {
    const double w = h[6] * x + h[7] * y + h[8];
    *ox = (h[0] * x + h[1] * y + h[2]) / w;
    *oy = (h[3] * x + h[4] * y + h[5]) / w;
}

// ---- plate carree (CAR) inputs: align_using_initial_carrington, alignment.py:344-399 --------------------
// wcslib celset (cel.c) for a cylindrical projection, fiducial native point (phi0, theta0) = (0, 0): celestial
// longitude / latitude of the native pole and LONPOLE, in degrees.  A CRVAL2 lag makes the projection oblique; an
// explicit LONPOLE on the wrong side of the equator leaves no valid pole (astropy raises InvalidTransformError, the
// reference's worker dies): returns 1, the caller reports NaN for that lag-point.
inline int car_euler(const coreg_wcs2d& w, ld* lngp_out, ld* latp_out, ld* phip_out) {
    const ld d2r = (ld)kPi / 180.0L, r2d = 180.0L / (ld)kPi;
    const ld lng0 = (ld)w.crval1 * (ld)w.unit_to_deg, lat0 = (ld)w.crval2 * (ld)w.unit_to_deg;
    const ld phi0 = 0, theta0 = 0;
    const ld phip = (w.lonpole == w.lonpole) ? (ld)w.lonpole : (lat0 >= theta0 ? 0.0L : 180.0L);
    const ld latpreq = (w.latpole == w.latpole) ? (ld)w.latpole : 90.0L;
    const ld tol = 1.0e-10L;
    const ld slat0 = sinl(lat0 * d2r), clat0 = cosl(lat0 * d2r);
    const ld cthe0 = 1.0L, sthe0 = 0.0L;
    ld sphip = 0, cphip = 1;
    if (phip != phi0) {
        sphip = sinl((phip - phi0) * d2r);
        cphip = cosl((phip - phi0) * d2r);
    }
    ld x = cthe0 * cphip, y = sthe0;
    ld z = hypotl(x, y), latp;
    if (z == 0) {
        if (slat0 != 0) return 1;
        latp = latpreq;
    } else {
        ld slz = slat0 / z;
        if (fabsl(slz) > 1) {
            if (fabsl(slz) - 1 < tol) slz = slz > 0 ? 1 : -1;
            else return 1;
        }
        const ld u = atan2l(y, x) * r2d, v = acosl(slz) * r2d;
        ld latp1 = u + v, latp2 = u - v;
        if (latp1 > 180) latp1 -= 360; else if (latp1 < -180) latp1 += 360;
        if (latp2 > 180) latp2 -= 360; else if (latp2 < -180) latp2 += 360;
        if (fabsl(latpreq - latp1) < fabsl(latpreq - latp2)) latp = fabsl(latp1) < 90 + tol ? latp1 : latp2;
        else latp = fabsl(latp2) < 90 + tol ? latp2 : latp1;
        if (!(fabsl(latp) < 90 + tol)) return 1;  // "No valid solution for latp"
        if (latp > 90) latp = 90; else if (latp < -90) latp = -90;
    }
    ld lngp;
    z = cosl(latp * d2r) * clat0;
    if (fabsl(z) < tol) {
        if (fabsl(clat0) < tol) lngp = lng0;
        else if (latp > 0) lngp = lng0 + phip - phi0 - 180;
        else lngp = lng0 - phip + phi0;
    } else {
        const ld xx = (sthe0 - sinl(latp * d2r) * slat0) / z, yy = sphip * cthe0 / clat0;
        if (xx == 0 && yy == 0) return 1;
        lngp = lng0 - atan2l(yy, xx) * r2d;
    }
    *lngp_out = lngp;
    *latp_out = latp;
    *phip_out = phip;
    return 0;
}
// native -> celestial rotation of a CAR header (same form as the zenithal one, with the pole from car_euler)
inline int car_native_to_celestial(const coreg_wcs2d& w, Mat3* out) {
    ld lngp, latp, phip;
    if (car_euler(w, &lngp, &latp, &phip)) return 1;
    const ld d2r = (ld)kPi / 180.0L;
    const ld ap = lngp * d2r, dp = latp * d2r, pp = phip * d2r;
    Mat3 rz1 = {{{cosl(ap), -sinl(ap), 0}, {sinl(ap), cosl(ap), 0}, {0, 0, 1}}};
    Mat3 t = {{{-sinl(dp), 0, cosl(dp)}, {0, -1, 0}, {cosl(dp), 0, sinl(dp)}}};
    Mat3 rz2 = {{{cosl(pp), sinl(pp), 0}, {-sinl(pp), cosl(pp), 0}, {0, 0, 1}}};
    *out = mat_mul(rz1, mat_mul(t, rz2));
    return 0;
}
// 0-based pixel -> native (phi, theta) [radians] of a CAR header: phi = x, theta = y (the intermediate world
// coordinates themselves), affine: (phi, theta) = A (i, j) + b.  And its inverse.
struct Affine2 {
    double m00, m01, m10, m11, b0, b1;
};
inline Affine2 car_pix_to_native(const coreg_wcs2d& w) {
    const Mat3 a = pix_to_iwc(w);
    return {(double)a.m[0][0], (double)a.m[0][1], (double)a.m[1][0], (double)a.m[1][1], (double)a.m[0][2],
            (double)a.m[1][2]};
}
inline Affine2 car_native_to_pix(const coreg_wcs2d& w) {
    const Mat3 a = iwc_to_pix(w);
    return {(double)a.m[0][0], (double)a.m[0][1], (double)a.m[1][0], (double)a.m[1][1], (double)a.m[0][2],
            (double)a.m[1][2]};
}
// The whole CAR -> CAR map of one lag-point, host version (planning, tests): pixel of `from` -> pixel of `to`.
//   n = unit vector of native (phi, theta) of `from`;  m = R n with R = R_to^T R_from;  (phi', theta') = (atan2, asin)
struct CarMapHost {
    Affine2 fwd, inv;
    double r[9];
    int init(const coreg_wcs2d& from, const coreg_wcs2d& to) {
        Mat3 rf, rt;
        if (car_native_to_celestial(from, &rf) || car_native_to_celestial(to, &rt)) return 1;
        const Mat3 m = mat_mul(mat_T(rt), rf);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r[3 * i + j] = (double)m.m[i][j];
        fwd = car_pix_to_native(from);
        inv = car_native_to_pix(to);
        return 0;
    }
    void apply(double x, double y, double* ox, double* oy) const {
        const double phi = fwd.m00 * x + fwd.m01 * y + fwd.b0, th = fwd.m10 * x + fwd.m11 * y + fwd.b1;
        const double ct = std::cos(th), n0 = ct * std::cos(phi), n1 = ct * std::sin(phi), n2 = std::sin(th);
        const double m0 = r[0] * n0 + r[1] * n1 + r[2] * n2, m1 = r[3] * n0 + r[4] * n1 + r[5] * n2,
                     m2 = r[6] * n0 + r[7] * n1 + r[8] * n2;
        const double p = std::atan2(m1, m0), t = std::asin(std::fmax(-1.0, std::fmin(1.0, m2)));
        *ox = inv.m00 * p + inv.m01 * t + inv.b0;
        *oy = inv.m10 * p + inv.m11 * t + inv.b1;
    }
};

// ---- Carrington ---------------------------------------------------------------------------------------
// Lag-independent-per-(roll, cdelt) part of utils/rectify.py:387-415 + :340-363.
struct CarrCommon {
    double dist;    // DSUN_OBS / (solar_r * R_sun)
    double cb, sb;  // cos / sin of radians(CRLT_OBS)
    double cr, sr;  // cos / sin of radians(roll)
    double cdelt1, cdelt2;
};
inline CarrCommon carr_common(const coreg_wcs2d& h, double solar_r) {
    CarrCommon c;
    c.dist = h.dsun_obs / (solar_r * kRSun);
    const double b0 = h.crlt_obs * kDeg2Rad;
    c.cb = std::cos(b0);
    c.sb = std::sin(b0);
    const double roll = h.crota * kDeg2Rad;
    c.cr = std::cos(roll);
    c.sr = std::sin(roll);
    c.cdelt1 = h.cdelt1;
    c.cdelt2 = h.cdelt2;
    return c;
}
// X0, Y0 of utils/rectify.py:399-404 (self.x, self.y of SphericalTransform)
inline void carr_origin(const coreg_wcs2d& h, double* x0, double* y0) {
    const double roll = h.crota * kDeg2Rad;
    const double c = std::cos(roll), s = std::sin(roll);
    const double dx = c * h.crval1 + s * h.crval2;
    const double dy = -s * h.crval1 + c * h.crval2;
    *x0 = (h.crpix1 - 1) - dx / h.cdelt1;
    *y0 = (h.crpix2 - 1) - dy / h.cdelt2;
}

// numpy.linspace(lo, hi, n, dtype=float32): float64 arithmetic (i*step then +lo, last := hi), cast at the end.
inline void linspace_f32(double lo, double hi, int n, std::vector<float>& out) {
    out.resize(n);
    if (n == 1) {
        out[0] = (float)lo;
        return;
    }
    const double div = (double)(n - 1);
    const double delta = hi - lo;
    const double step = delta / div;
    for (int i = 0; i < n; ++i) {
        volatile double y = (step == 0.0) ? ((double)i / div) * delta : (double)i * step;  // no fma with +lo
        out[i] = (float)(y + lo);
    }
    out[n - 1] = (float)hi;
}

// Per-column / per-row trig tables of the Carrington grid (the only transcendental inputs of
// SphericalTransform.forward, utils/rectify.py:342-347):
//   lon' = radians(float64(lon32)) - radians(CRLN_OBS)            -> sin, cos in float64
//   lat  = radians(lat32) in float32                               -> sin, cos in float32 (quirk Q6)
struct CarrTables {
    std::vector<double> sin_lon, cos_lon;  // [n_lon]
    std::vector<float> cos_lat, sin_lat;   // [n_lat]
};
inline void carr_tables(const coreg_carr_grid& g, double crln_obs, CarrTables& t) {
    std::vector<float> lon32, lat32;
    linspace_f32(g.lon0, g.lon1, g.n_lon, lon32);
    linspace_f32(g.lat0, g.lat1, g.n_lat, lat32);
    const double l0 = crln_obs * kDeg2Rad;
    t.sin_lon.resize(g.n_lon);
    t.cos_lon.resize(g.n_lon);
    for (int i = 0; i < g.n_lon; ++i) {
        const double lon = (double)lon32[i] * kDeg2Rad - l0;
        t.sin_lon[i] = std::sin(lon);
        t.cos_lon[i] = std::cos(lon);
    }
    t.cos_lat.resize(g.n_lat);
    t.sin_lat.resize(g.n_lat);
    const float d2r32 = (float)kDeg2Rad;
    for (int j = 0; j < g.n_lat; ++j) {
        volatile float latr = lat32[j] * d2r32;  // float32 multiply, as numpy.radians on a float32 array
        t.cos_lat[j] = g.lat_cos ? g.lat_cos[j] : (float)std::cos((double)latr);
        t.sin_lat[j] = g.lat_sin ? g.lat_sin[j] : (float)std::sin((double)latr);
    }
}

}  // namespace coreg
