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
#include <cstdlib>
#include <utility>
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
    // a lag that leaves no header to evaluate (CDELT + d = 0, a non-finite card): the lag-point stays NaN, as one whose
    // worker died in the reference (astropy refuses such a header)
    if (!(out->cdelt1 != 0.0 && out->cdelt2 != 0.0) || !std::isfinite(out->cdelt1) || !std::isfinite(out->cdelt2) ||
        !std::isfinite(out->crota) || (change_pcij && (!std::isfinite(out->pc1_2) || !std::isfinite(out->pc2_1))))
        return 2;
    return 0;
}

// What the sweeps, resamplers and reference preparations refuse (COREG_EINVAL) before anything is planned or launched:
// a header the arithmetic below cannot turn into finite pixel coordinates.  `carrington_transform`: the header feeds
// utils/rectify.py's transform (CRPIX, CRVAL, CDELT, CROTA, DSUN_OBS, CRLN_OBS, CRLT_OBS; PCi_j ignored, quirk Q4);
// otherwise it is a TAN or CAR WCS (PCi_j, unit, LONPOLE; LATPOLE and a CAR LONPOLE may be NaN = FITS default).
inline const char* wcs_problem(const coreg_wcs2d& w, bool carrington_transform) {
    const double core[] = {w.crpix1, w.crpix2, w.crval1, w.crval2, w.cdelt1, w.cdelt2, w.crota};
    for (double v : core)
        if (!std::isfinite(v)) return "a non-finite CRPIX / CRVAL / CDELT / CROTA card";
    if (w.cdelt1 == 0.0 || w.cdelt2 == 0.0) return "CDELT1 or CDELT2 is zero";
    // (finite but absurd cards overflow to Inf, then Inf - Inf, a few products later: same refusal)
    for (double v : core)
        if (std::fabs(v) > 1e12) return "a CRPIX / CRVAL / CDELT / CROTA card beyond 1e12 in magnitude";
    if (std::fabs(w.cdelt1) < 1e-30 || std::fabs(w.cdelt2) < 1e-30) return "CDELT1 or CDELT2 below 1e-30 in magnitude";
    if (carrington_transform) {
        if (!std::isfinite(w.dsun_obs) || !(w.dsun_obs > 0.0) || w.dsun_obs > 1e18)
            return "DSUN_OBS must be a positive distance (metres, below 1e18)";
        if (!std::isfinite(w.crln_obs) || !std::isfinite(w.crlt_obs)) return "a non-finite CRLN_OBS / CRLT_OBS card";
        return nullptr;
    }
    const double pc[] = {w.pc1_1, w.pc1_2, w.pc2_1, w.pc2_2};
    for (double v : pc)
        if (!std::isfinite(v)) return "a non-finite PCi_j card";
    for (double v : pc)
        if (std::fabs(v) > 1e12) return "a PCi_j card beyond 1e12 in magnitude";
    if (!std::isfinite(w.unit_to_deg) || !(w.unit_to_deg > 0.0) || w.unit_to_deg > 1e6)
        return "unit_to_deg must be positive (and at most 1e6)";
    const double det = w.pc1_1 * w.pc2_2 - w.pc1_2 * w.pc2_1;
    if (!(det != 0.0) || !std::isfinite(det)) return "PCi_j is singular";
    if (w.proj != COREG_PROJ_CAR && !std::isfinite(w.lonpole)) return "a non-finite LONPOLE card";
    if (std::isinf(w.lonpole) || std::isinf(w.latpole)) return "an infinite LONPOLE / LATPOLE card";
    return nullptr;
}
inline const char* grid_problem(const coreg_carr_grid& g) {
    if (g.n_lon < 1 || g.n_lat < 1) return "Carrington grid: n_lon and n_lat must be at least 1";
    if (!std::isfinite(g.lon0) || !std::isfinite(g.lon1) || !std::isfinite(g.lat0) || !std::isfinite(g.lat1) ||
        std::fabs(g.lon0) > 1e6 || std::fabs(g.lon1) > 1e6 || std::fabs(g.lat0) > 1e6 || std::fabs(g.lat1) > 1e6)
        return "Carrington grid: non-finite (or absurd) limits";
    return nullptr;
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
    // P[i2] * Q[i1], filled on first use: a sweep over several (cdelt, crota) combinations asks for every (i1, i2) once per
    // combination (same product, same bits)
    mutable std::vector<Mat3d> PQ;
    mutable std::vector<unsigned char> have_pq;
    // from: header whose pixel grid is mapped (target grid); ref: unshifted header being lagged
    void init(const coreg_wcs2d& from, const coreg_wcs2d& ref, const double* lag1, int n1, const double* lag2,
              int n2, bool cache_products = false) {
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
        PQ.clear();
        have_pq.clear();
        if (cache_products && (size_t)n1 * n2 <= ((size_t)1 << 20)) {  // (72 B per lag of the plane)
            PQ.assign((size_t)n1 * n2, Mat3d());
            have_pq.assign((size_t)n1 * n2, 0);
        }
    }
    // every product of the cache at once: afterwards `get` only reads (several threads may call it)
    void fill_products(int i1_lo, int i1_hi) const {  // (the CRVAL1 lags of the sweep's slice, inclusive)
        if (PQ.empty()) return;
        for (size_t i2 = 0; i2 < P.size(); ++i2)
            for (size_t i1 = (size_t)i1_lo; i1 <= (size_t)i1_hi && i1 < Q.size(); ++i1) {
                const size_t k = i2 * Q.size() + i1;
                if (!have_pq[k]) {
                    PQ[k] = mul3(P[i2], Q[i1]);
                    have_pq[k] = 1;
                }
            }
    }
    // per (cdelt, crota) combination: B = iwc_to_pix(shifted header) * Ninv
    static Mat3d combo(const coreg_wcs2d& shifted) {
        const Mat3 Ninv = {{{0, 1, 0}, {-1, 0, 0}, {0, 0, 1}}};
        return to_double(mat_mul(iwc_to_pix(shifted), Ninv));
    }
    void get(const Mat3d& B, int i1, int i2, double h[9]) const {
        Mat3d pq;
        if (!PQ.empty()) {
            const size_t k = (size_t)i2 * Q.size() + i1;
            if (!have_pq[k]) {
                PQ[k] = mul3(P[i2], Q[i1]);
                have_pq[k] = 1;
            }
            pq = PQ[k];
        } else {
            pq = mul3(P[i2], Q[i1]);
        }
        const Mat3d m = mul3(B, pq);
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

// ---- wcslib's own pixel -> sky -> pixel chain, operation by operation -------------------------------------------
// For ONE lag-point of a helioprojective sweep the exact homography is not what decides the result: the zero lag of the
// parallelism=True path, where the target header IS the shifted header (alignment.py:1000, :1038-1069).  The map is
// then the identity up to wcslib's rounding noise (|eps| ~ 1e-12..1e-9 px), and the sign of that noise decides,
// through the bounds rule c < 0 or c > n-1 (utils/Util.py:98-102 -> scipy map_coordinates), whether a border pixel
// of the grid is kept.  Reproducing the decision needs wcslib's arithmetic itself.  Restated from wcslib 7.x
// (third-party, bundled with astropy; absent from the reference tree): lin.c linp2x / linx2p / matinv, prj.c tanx2s /
// tans2x, sph.c sphx2s / sphs2x, wcstrig.c -- including the macro expansions that fix the order of the roundings
// (`#define D2R PI/180.0`, `#define R2D 180.0/PI`: angle*D2R is (angle*PI)/180, atan2(y,x)*R2D is (atan2*180)/PI) and
// glibc's sincos().  Pinned bit for bit, for every border pixel of five headers, against astropy 4.3.1 / wcslib 7.6
// (tests/golden/border_golden.npz).  The numbers depend on the libm in use, as the reference's do.
struct WcslibTan {
    double crpix[2], cdelt[2], piximg[2][2], imgpix[2][2];
    bool unity;
    double e0, e1, e2, e3, e4;  // celprm euler: lng_p, 90 - lat_p, phi_p, cos(e1), sin(e1)

    // wcstrig.c: abs((int)floor(v)) % 4 -- through fmod, so that an angle of 1e20 degrees in a header is not an
    // out-of-range float -> int conversion (same value wherever wcslib's own cast is defined)
    static int quadrant(double v) { return (int)std::fmod(std::fabs(std::floor(v)), 4.0); }
    static void sincosd(double a, double* s, double* c) {
#pragma clang fp contract(off)
        if (std::fmod(a, 90.0) == 0.0) {
            const int i = quadrant(a / 90.0 + 0.5);
            switch (i) {
                case 0: *s = 0.0; *c = 1.0; return;
                case 1: *s = (a > 0.0) ? 1.0 : -1.0; *c = 0.0; return;
                case 2: *s = 0.0; *c = -1.0; return;
                default: *s = (a > 0.0) ? -1.0 : 1.0; *c = 0.0; return;
            }
        }
        ::sincos(a * kPi / 180.0, s, c);
    }
    static double cosd(double a) {
#pragma clang fp contract(off)
        if (std::fmod(a, 90.0) == 0.0) {
            const int i = quadrant(a / 90.0 + 0.5);
            return i == 0 ? 1.0 : (i == 2 ? -1.0 : 0.0);
        }
        return std::cos(a * kPi / 180.0);
    }
    static double sind(double a) {
#pragma clang fp contract(off)
        if (std::fmod(a, 90.0) == 0.0) {
            const int i = quadrant(a / 90.0 - 0.5);
            return i == 0 ? 1.0 : (i == 2 ? -1.0 : 0.0);
        }
        return std::sin(a * kPi / 180.0);
    }
    static double atan2d(double y, double x) {
#pragma clang fp contract(off)
        if (y == 0.0) {
            if (x >= 0.0) return 0.0;
            if (x < 0.0) return 180.0;
        } else if (x == 0.0) {
            if (y > 0.0) return 90.0;
            if (y < 0.0) return -90.0;
        }
        return std::atan2(y, x) * 180.0 / kPi;
    }
    static double asind(double v) {
#pragma clang fp contract(off)
        if (v <= -1.0) {
            if (v + 1.0 > -1e-10) return -90.0;
        } else if (v == 0.0) {
            return 0.0;
        } else if (v >= 1.0) {
            if (v - 1.0 < 1e-10) return 90.0;
        }
        return std::asin(v) * 180.0 / kPi;
    }
    static double acosd(double v) {
#pragma clang fp contract(off)
        if (v >= 1.0) {
            if (v - 1.0 < 1e-10) return 0.0;
        } else if (v == 0.0) {
            return 90.0;
        } else if (v <= -1.0) {
            if (v + 1.0 > -1e-10) return 180.0;
        }
        return std::acos(v) * 180.0 / kPi;
    }
    // lin.c matinv() for n = 2: LU decomposition with scaled partial pivoting, then one solve per unit vector
    static void matinv2(const double m[2][2], double inv[2][2]) {
#pragma clang fp contract(off)
        int mxl[2] = {0, 1}, lxm[2] = {0, 0};
        double rowmax[2] = {0.0, 0.0}, lu[2][2] = {{m[0][0], m[0][1]}, {m[1][0], m[1][1]}};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) rowmax[i] = std::fmax(rowmax[i], std::fabs(m[i][j]));
        for (int k = 0; k < 2; ++k) {
            double colmax = std::fabs(lu[k][k]) / rowmax[k];
            int pivot = k;
            for (int i = k + 1; i < 2; ++i) {
                const double d = std::fabs(lu[i][k]) / rowmax[i];
                if (d > colmax) {
                    colmax = d;
                    pivot = i;
                }
            }
            if (pivot > k) {
                for (int j = 0; j < 2; ++j) std::swap(lu[pivot][j], lu[k][j]);
                std::swap(rowmax[pivot], rowmax[k]);
                std::swap(mxl[pivot], mxl[k]);
            }
            for (int i = k + 1; i < 2; ++i)
                if (lu[i][k] != 0.0) {
                    lu[i][k] /= lu[k][k];
                    for (int j = k + 1; j < 2; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
                }
        }
        for (int i = 0; i < 2; ++i) lxm[mxl[i]] = i;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) inv[i][j] = 0.0;
        for (int k = 0; k < 2; ++k) {
            inv[lxm[k]][k] = 1.0;
            for (int i = lxm[k] + 1; i < 2; ++i)
                for (int j = lxm[k]; j < i; ++j) inv[i][k] -= lu[i][j] * inv[j][k];
            for (int i = 1; i >= 0; --i) {
                for (int j = i + 1; j < 2; ++j) inv[i][k] -= lu[i][j] * inv[j][k];
                inv[i][k] /= lu[i][i];
            }
        }
    }
    void init(const coreg_wcs2d& w) {
#pragma clang fp contract(off)
        crpix[0] = w.crpix1;
        crpix[1] = w.crpix2;
        // wcslib's unit fix scales CDELT and CRVAL to degrees (arcsec: x 1/3600)
        cdelt[0] = w.cdelt1 * w.unit_to_deg;
        cdelt[1] = w.cdelt2 * w.unit_to_deg;
        unity = w.pc1_1 == 1.0 && w.pc2_2 == 1.0 && w.pc1_2 == 0.0 && w.pc2_1 == 0.0;
        piximg[0][0] = cdelt[0] * w.pc1_1;
        piximg[0][1] = cdelt[0] * w.pc1_2;
        piximg[1][0] = cdelt[1] * w.pc2_1;
        piximg[1][1] = cdelt[1] * w.pc2_2;
        matinv2(piximg, imgpix);
        e0 = w.crval1 * w.unit_to_deg;  // zenithal projection: the pole of the native system is the reference point
        e1 = 90.0 - w.crval2 * w.unit_to_deg;
        e2 = w.lonpole;
        sincosd(e1, &e4, &e3);
    }
    // wcsp2s: 0-based pixel -> (lng, lat) degrees
    void p2s(double px0, double py0, double* lng_out, double* lat_out) const {
#pragma clang fp contract(off)
        const double t0 = (px0 + 1.0) - crpix[0], t1 = (py0 + 1.0) - crpix[1];
        double x, y;
        if (unity) {
            x = cdelt[0] * t0;
            y = cdelt[1] * t1;
        } else {
            x = 0.0;
            y = 0.0;
            x += piximg[0][0] * t0;
            y += piximg[1][0] * t0;
            x += piximg[0][1] * t1;
            y += piximg[1][1] * t1;
        }
        const double xj = x + 0.0, yj = y + 0.0;
        const double r = std::sqrt(xj * xj + yj * yj);
        const double phi = r == 0.0 ? 0.0 : atan2d(xj, -yj);
        const double theta = atan2d(180.0 / kPi, r);
        const double dphi = phi - e2;
        double sinthe, costhe, sinphi, cosphi;
        sincosd(theta, &sinthe, &costhe);
        const double costhe3 = costhe * e3, costhe4 = costhe * e4, sinthe3 = sinthe * e3, sinthe4 = sinthe * e4;
        sincosd(dphi, &sinphi, &cosphi);
        double xx = sinthe4 - costhe3 * cosphi;
        if (std::fabs(xx) < 1.0e-5) xx = -cosd(theta + e1) + costhe3 * (1.0 - cosphi);
        const double yy = -costhe * sinphi;
        const double dlng = (xx != 0.0 || yy != 0.0) ? atan2d(yy, xx) : dphi + 180.0;
        double lng = e0 + dlng;
        if (e0 >= 0.0) {
            if (lng < 0.0) lng += 360.0;
        } else {
            if (lng > 0.0) lng -= 360.0;
        }
        if (lng > 360.0) lng -= 360.0;
        else if (lng < -360.0) lng += 360.0;
        double lat;
        if (std::fmod(dphi, 180.0) == 0.0) {
            lat = theta + cosphi * e1;
            if (lat > 90.0) lat = 180.0 - lat;
            if (lat < -90.0) lat = -180.0 - lat;
        } else {
            const double z = sinthe3 + costhe4 * cosphi;
            if (std::fabs(z) > 0.99) lat = std::copysign(acosd(std::sqrt(xx * xx + yy * yy)), z);
            else lat = asind(z);
        }
        *lng_out = lng;
        *lat_out = lat;
    }
    // wcss2p: (lng, lat) degrees -> 0-based pixel (NaN where wcslib flags the point)
    void s2p(double lng, double lat, double* px_out, double* py_out) const {
#pragma clang fp contract(off)
        const double dlng = lng - e0;
        double sinlat, coslat, sinlng, coslng;
        sincosd(lat, &sinlat, &coslat);
        const double coslat3 = coslat * e3, coslat4 = coslat * e4, sinlat3 = sinlat * e3, sinlat4 = sinlat * e4;
        sincosd(dlng, &sinlng, &coslng);
        double xx = sinlat4 - coslat3 * coslng;
        if (std::fabs(xx) < 1.0e-5) xx = -cosd(lat + e1) + coslat3 * (1.0 - coslng);
        const double yy = -coslat * sinlng;
        const double dphi = (xx != 0.0 || yy != 0.0) ? atan2d(yy, xx) : dlng - 180.0;
        double phi = std::fmod(e2 + dphi, 360.0);
        if (phi > 180.0) phi -= 360.0;
        else if (phi < -180.0) phi += 360.0;
        double theta;
        if (std::fmod(dlng, 180.0) == 0.0) {
            theta = lat + coslng * e1;
            if (theta > 90.0) theta = 180.0 - theta;
            if (theta < -90.0) theta = -180.0 - theta;
        } else {
            const double z = sinlat3 + coslat4 * coslng;
            if (std::fabs(z) > 0.99) theta = std::copysign(acosd(std::sqrt(xx * xx + yy * yy)), z);
            else theta = asind(z);
        }
        double sinphi, cosphi;
        sincosd(phi, &sinphi, &cosphi);
        const double s = sind(theta);
        if (s == 0.0 || s < 0.0) {  // tans2x: divergent / behind the tangent plane -> invalid
            *px_out = *py_out = std::nan("");
            return;
        }
        const double r = 180.0 / kPi * cosd(theta) / s;
        const double x = r * sinphi - 0.0, y = -r * cosphi - 0.0;
        double p0, p1;
        if (unity) {
            p0 = x / cdelt[0] + crpix[0];
            p1 = y / cdelt[1] + crpix[1];
        } else {
            p0 = 0.0;
            p0 += imgpix[0][0] * x;
            p0 += imgpix[0][1] * y;
            p0 += crpix[0];
            p1 = 0.0;
            p1 += imgpix[1][0] * x;
            p1 += imgpix[1][1] * y;
            p1 += crpix[1];
        }
        *px_out = p0 - 1.0;
        *py_out = p1 - 1.0;
    }
};
// AlignCommonUtil.ang2pipi on a value in degrees (utils/Util.py:76-80; numpy's remainder takes the divisor's sign)
inline double ang2pipi_deg(double a) {
#pragma clang fp contract(off)
    const double x = -a + 180.0;
    double m = std::fmod(x, 360.0);
    if (m != 0.0) {
        if (m < 0.0) m += 360.0;
    } else {
        m = 0.0;
    }
    return -(m - 180.0);
}
// alignment.py:1038-1069 for one pixel of the target header: pixel -> sky (`from`) -> ang2pipi -> pixel (`to`)
inline void wcslib_pixel_to_pixel(const WcslibTan& from, const WcslibTan& to, double px, double py, double* ox,
                                  double* oy) {
    double lng, lat;
    from.p2s(px, py, &lng, &lat);
    to.s2p(ang2pipi_deg(lng), ang2pipi_deg(lat), ox, oy);
}

// The same chain for a plate-carree header (CTYPE -CAR; two Carrington maps, align_using_initial_carrington,
// alignment.py:344-399 -- there the reference applies NO ang2pipi: lon_ctype "CRLN-CAR", utils/Util.py:302-304):
// wcslib's cel.c celset() for a cylindrical projection (fiducial native point (0, 0)) in plain double, operation by
// operation; prj.c carx2s / cars2x (default r0: both scale factors exactly 1); sph.c sphx2s / sphs2x INCLUDING the
// "simple change in origin of longitude" branch an equatorial map takes.  One measured oddity (southern maps, LONPOLE =
// 180): the sine entering the longitude of the native pole is libm's sin(pi) = 1.22e-16, not wcstrig's exact 0.  Pinned bit
// for bit by tests/golden/border_car_golden.npz (astropy 4.3.1 / wcslib 7.6, nine headers, every border pixel).
struct WcslibCar : WcslibTan {
    bool valid = true;
    void init(const coreg_wcs2d& w) {
#pragma clang fp contract(off)
        WcslibTan::init(w);
        const double tol = 1.0e-10;
        const double lng0 = w.crval1 * w.unit_to_deg, lat0 = w.crval2 * w.unit_to_deg;
        const double phi0 = 0.0, theta0 = 0.0;
        double latp = (w.latpole == w.latpole) ? w.latpole : 90.0;
        double phip;
        if (!(w.lonpole == w.lonpole) || w.lonpole == 999.0) {
            phip = (lat0 < theta0) ? 180.0 : 0.0;
            phip += phi0;
            if (phip < -180.0) phip += 360.0;
            else if (phip > 180.0) phip -= 360.0;
        } else {
            phip = w.lonpole;
        }
        double slat0, clat0, sthe0, cthe0, sphip, cphip, u = 0.0, v = 0.0;
        sincosd(lat0, &slat0, &clat0);
        sincosd(theta0, &sthe0, &cthe0);
        int latpreq = 0;
        valid = true;
        if (phip == phi0) {
            sphip = 0.0;
            cphip = 1.0;
            u = theta0;
            v = 90.0 - lat0;
        } else {
            ::sincos((phip - phi0) * kPi / 180.0, &sphip, &cphip);
            const double x = cthe0 * cphip, y = sthe0, z = std::sqrt(x * x + y * y);
            if (z == 0.0) {
                if (slat0 != 0.0) valid = false;
                latpreq = 2;
                if (latp > 90.0) latp = 90.0;
                else if (latp < -90.0) latp = -90.0;
            } else {
                double slz = slat0 / z;
                if (std::fabs(slz) > 1.0) {
                    if ((std::fabs(slz) - 1.0) < tol) slz = (slz > 0.0) ? 1.0 : -1.0;
                    else valid = false;
                }
                u = atan2d(y, x);
                v = acosd(slz);
            }
        }
        if (latpreq == 0) {
            double latp1 = u + v;
            if (latp1 > 180.0) latp1 -= 360.0;
            else if (latp1 < -180.0) latp1 += 360.0;
            double latp2 = u - v;
            if (latp2 > 180.0) latp2 -= 360.0;
            else if (latp2 < -180.0) latp2 += 360.0;
            if (std::fabs(latp - latp1) < std::fabs(latp - latp2)) latp = (std::fabs(latp1) < 90.0 + tol) ? latp1 : latp2;
            else latp = (std::fabs(latp2) < 90.0 + tol) ? latp2 : latp1;
            if (std::fabs(latp) < 90.0 + tol) {
                if (latp > 90.0) latp = 90.0;
                else if (latp < -90.0) latp = -90.0;
            } else {
                valid = false;
            }
        }
        double lngp;
        const double z = cosd(latp) * clat0;
        if (std::fabs(z) < tol) {
            if (std::fabs(clat0) < tol) lngp = lng0;
            else if (latp > 0.0) lngp = lng0 + phip - phi0 - 180.0;
            else lngp = lng0 - phip + phi0;
        } else {
            const double x = (sthe0 - sind(latp) * slat0) / z, y = sphip * cthe0 / clat0;
            if (x == 0.0 && y == 0.0) valid = false;
            lngp = lng0 - atan2d(y, x);
        }
        if (lng0 >= 0.0) {
            if (lngp < 0.0) lngp += 360.0;
            else if (lngp > 360.0) lngp -= 360.0;
        } else {
            if (lngp > 0.0) lngp -= 360.0;
            else if (lngp < -360.0) lngp += 360.0;
        }
        e0 = lngp;
        e1 = 90.0 - latp;
        e2 = phip;
        sincosd(e1, &e4, &e3);
    }
    void p2s(double px0, double py0, double* lng_out, double* lat_out) const {
#pragma clang fp contract(off)
        const double t0 = (px0 + 1.0) - crpix[0], t1 = (py0 + 1.0) - crpix[1];
        double x, y;
        if (unity) {
            x = cdelt[0] * t0;
            y = cdelt[1] * t1;
        } else {
            x = 0.0;
            y = 0.0;
            x += piximg[0][0] * t0;
            y += piximg[1][0] * t0;
            x += piximg[0][1] * t1;
            y += piximg[1][1] * t1;
        }
        const double phi = 1.0 * (x + 0.0), theta = 1.0 * (y + 0.0);  // carx2s
        double lng, lat;
        if (e4 == 0.0) {
            if (e1 == 0.0) {
                const double dlng = std::fmod(e0 + 180.0 - e2, 360.0);
                lng = phi + dlng;
                lat = theta;
            } else {
                const double dlng = std::fmod(e0 + e2, 360.0);
                lng = dlng - phi;
                lat = -theta;
            }
            if (e0 >= 0.0) {
                if (lng < 0.0) lng += 360.0;
            } else {
                if (lng > 0.0) lng -= 360.0;
            }
            if (lng > 360.0) lng -= 360.0;
            else if (lng < -360.0) lng += 360.0;
            *lng_out = lng;
            *lat_out = lat;
            return;
        }
        const double dphi = phi - e2;
        double sinthe, costhe, sinphi, cosphi;
        sincosd(theta, &sinthe, &costhe);
        const double costhe3 = costhe * e3, costhe4 = costhe * e4, sinthe3 = sinthe * e3, sinthe4 = sinthe * e4;
        sincosd(dphi, &sinphi, &cosphi);
        double xx = sinthe4 - costhe3 * cosphi;
        if (std::fabs(xx) < 1.0e-5) xx = -cosd(theta + e1) + costhe3 * (1.0 - cosphi);
        const double yy = -costhe * sinphi;
        const double dlng = (xx != 0.0 || yy != 0.0) ? atan2d(yy, xx) : dphi + 180.0;
        lng = e0 + dlng;
        if (e0 >= 0.0) {
            if (lng < 0.0) lng += 360.0;
        } else {
            if (lng > 0.0) lng -= 360.0;
        }
        if (lng > 360.0) lng -= 360.0;
        else if (lng < -360.0) lng += 360.0;
        if (std::fmod(dphi, 180.0) == 0.0) {
            lat = theta + cosphi * e1;
            if (lat > 90.0) lat = 180.0 - lat;
            if (lat < -90.0) lat = -180.0 - lat;
        } else {
            const double z = sinthe3 + costhe4 * cosphi;
            if (std::fabs(z) > 0.99) lat = std::copysign(acosd(std::sqrt(xx * xx + yy * yy)), z);
            else lat = asind(z);
        }
        *lng_out = lng;
        *lat_out = lat;
    }
    void s2p(double lng, double lat, double* px_out, double* py_out) const {
#pragma clang fp contract(off)
        double phi, theta;
        if (e4 == 0.0) {
            if (e1 == 0.0) {
                const double dphi = std::fmod(e2 - 180.0 - e0, 360.0);
                phi = std::fmod(lng + dphi, 360.0);
                theta = lat;
            } else {
                const double dphi = std::fmod(e2 + e0, 360.0);
                phi = std::fmod(dphi - lng, 360.0);
                theta = -lat;
            }
            if (phi > 180.0) phi -= 360.0;
            else if (phi < -180.0) phi += 360.0;
        } else {
            const double dlng = lng - e0;
            double sinlat, coslat, sinlng, coslng;
            sincosd(lat, &sinlat, &coslat);
            const double coslat3 = coslat * e3, coslat4 = coslat * e4, sinlat3 = sinlat * e3, sinlat4 = sinlat * e4;
            sincosd(dlng, &sinlng, &coslng);
            double xx = sinlat4 - coslat3 * coslng;
            if (std::fabs(xx) < 1.0e-5) xx = -cosd(lat + e1) + coslat3 * (1.0 - coslng);
            const double yy = -coslat * sinlng;
            const double dphi = (xx != 0.0 || yy != 0.0) ? atan2d(yy, xx) : dlng - 180.0;
            phi = std::fmod(e2 + dphi, 360.0);
            if (phi > 180.0) phi -= 360.0;
            else if (phi < -180.0) phi += 360.0;
            if (std::fmod(dlng, 180.0) == 0.0) {
                theta = lat + coslng * e1;
                if (theta > 90.0) theta = 180.0 - theta;
                if (theta < -90.0) theta = -180.0 - theta;
            } else {
                const double z = sinlat3 + coslat4 * coslng;
                if (std::fabs(z) > 0.99) theta = std::copysign(acosd(std::sqrt(xx * xx + yy * yy)), z);
                else theta = asind(z);
            }
        }
        const double x = 1.0 * phi - 0.0, y = 1.0 * theta - 0.0;  // cars2x
        double p0, p1;
        if (unity) {
            p0 = x / cdelt[0] + crpix[0];
            p1 = y / cdelt[1] + crpix[1];
        } else {
            p0 = 0.0;
            p0 += imgpix[0][0] * x;
            p0 += imgpix[0][1] * y;
            p0 += crpix[0];
            p1 = 0.0;
            p1 += imgpix[1][0] * x;
            p1 += imgpix[1][1] * y;
            p1 += crpix[1];
        }
        *px_out = p0 - 1.0;
        *py_out = p1 - 1.0;
    }
};
inline void wcslib_pixel_to_pixel(const WcslibCar& from, const WcslibCar& to, double px, double py, double* ox, double* oy) {
    double lng, lat;
    from.p2s(px, py, &lng, &lat);
    to.s2p(lng, lat, ox, oy);  // (no ang2pipi on this path)
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

// Inputs of kernels.hpp car_tile_margin for one launch of the plate-carree sweep (tile = tile_w x kTilePts / tile_w
// target pixels): (tile half-diagonal in radians)^2 / 2 x pixels per radian of the shifted map; and the angle between
// the native poles of the two maps from the rotation R = R_shifted^T R_target (its [2][2] element).
inline double car_box_c(const coreg_wcs2d& target, const coreg_wcs2d& shifted, int tile_w, int tile_pts = 1024) {
    const double step = std::max(std::fabs(target.cdelt1), std::fabs(target.cdelt2)) * target.unit_to_deg * kDeg2Rad;
    const double s_half = 0.5 * step * std::hypot((double)tile_w, (double)(tile_pts / tile_w));
    const double px_per_rad =
        1.0 / (std::min(std::fabs(shifted.cdelt1), std::fabs(shifted.cdelt2)) * shifted.unit_to_deg * kDeg2Rad);
    return 0.5 * s_half * s_half * px_per_rad;
}
inline double car_pole_sep(const double r[9]) { return std::acos(std::fmax(-1.0, std::fmin(1.0, r[8]))); }

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
