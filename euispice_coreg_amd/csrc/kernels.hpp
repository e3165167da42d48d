// HIP kernels (gfx950 / CDNA4, wave64) of the alignment sweep.
//
// Work decomposition ("lag-per-lane"):
//   * the target grid is cut in tiles of 1024 points; a precompute pass (k_precompute) evaluates the
//     lag-independent part of the pixel-coordinate transform once per tile, drops points that can never
//     contribute (reference NaN, behind the limb, outside the small image for every lag) and stores the
//     survivors compacted, tile-major, together with their bounding box in small-image pixel space;
//   * the sweep kernel (k_sweep) gives every LANE one lag-point (a workgroup = 4 point-groups x 256 lags of a compact
//     CRVAL patch, all sharing one LDS window) and walks the compacted points of its tiles: point data are wave-uniform (scalar loads), every lane adds
//     its own lag displacement, gathers the 3x3 (order 2) / 2x2 (order 1) taps from an LDS-staged window of the
//     small image and accumulates its own six Pearson sums in registers.  No cross-lane reduction exists
//     anywhere on the hot path; partial sums leave the kernel once per (tile-group, lag);
//   * k_finalize adds the tile-group slabs in a fixed order (deterministic) and evaluates the coefficient.
//
// Reference arithmetic restated (paths relative to euispice_coreg/):
//   utils/Util.py:82-104 + scipy.ndimage.map_coordinates(order, mode='constant', prefilter=False)  -> spline_*()
//   utils/rectify.py:340-363 SphericalTransform.forward                                             -> carr_term()
//   hdrshift/alignment.py:525-531 mask + hdrshift/c_correlate.py:39-72 Pearson                      -> k_sweep/k_finalize
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace coreg {

#ifndef COREG_TILE_PTS
#define COREG_TILE_PTS 1024
#endif
#ifndef COREG_POINT_GROUPS
#define COREG_POINT_GROUPS 4
#endif
constexpr int kTilePts = COREG_TILE_PTS;  // grid points per tile
constexpr int kBlock = 256;     // lag slots per batch (one lag per lane, 4 waves)
constexpr int kPointGroups = COREG_POINT_GROUPS; // a sweep workgroup = kPointGroups x kBlock threads sharing one LDS window
constexpr int kSweepThreads = kBlock * kPointGroups;
constexpr int kChunk = 4;       // points per scalar-load chunk
constexpr int kNumSums = 6;     // n, sum a, sum b, sum aa, sum bb, sum ab

// one compacted grid point: lag-independent base coordinates + (reference value - pivot)
// pad: method 'residus': 1/sqrt(reference); MODE_CAR: third component of the unit vector; otherwise the sums of the
// point's chunk for the all-finite interior visits of k_sweep (point 4c: sum of the chunk's four a, point 4c + 1: sum of
// their squares, k_precompute)
struct __attribute__((aligned(32))) Pt {
    double b0, b1, a, pad;
};

// MODE_HOMOGRAPHY_SERIES: same map, denominator 1 + eps inverted as 1 - eps + eps^2 (host guarantees |eps| < 4e-6, i.e.
// a truncation error below 1e-16 relative); MODE_HOMOGRAPHY divides exactly (any field of view)
// MODE_CAR: plate-carree maps on both sides (align_using_initial_carrington): base coordinates = native (phi, theta) of
// the target pixel [radians], per lag a rotation of the sphere (h[0..8]) between the two native frames, then
// (atan2, asin) and the affine native -> pixel map of the shifted header (uniform per launch, LaunchU)
enum { MODE_TRANSLATE = 0, MODE_HOMOGRAPHY = 1, MODE_HOMOGRAPHY_SERIES = 2, MODE_CAR = 3 };
// per-launch uniforms of the coordinate map / sampler that are not per-lane
struct LaunchU {
    double m00, m01, m10, m11, b0, b1;  // MODE_CAR: native (phi, theta) [rad] -> 0-based pixel
    int order_rt;                       // ORDER == ORDER_RT kernels: the spline order (0..5)
    int h_incr;                         // HOMOGRAPHY[_SERIES], order 2, interior LDS visits: advance along runs (tile_points)
    double box_c;     // MODE_CAR: (tile half-diagonal [rad])^2 / 2 x pixels per radian of the shifted map (car_tile_margin)
    double pole_sep;  // MODE_CAR: largest angle [rad] between the native poles of the target and of a shifted map
};

// MODE_CAR: pixels by which the image of a tile may leave the bounding box of its four mapped corners.  The map
// (phi, theta) -> unit vector -> rotation -> (atan2, asin) -> pixel is not projective; over a tile of half-diagonal s
// [rad] a component f of it leaves the box of the corner values by at most sup|D^2 f| s^2 / 2, and on the sphere the
// second derivatives of the longitude grow like 1 / cos^2(latitude) towards the pole of the frame they are taken in
// (a great circle passing at distance d from a pole turns its longitude by pi within ~d).  th_abs = the largest
// |native latitude| the tile can reach in EITHER frame (its own extent plus the angle between the two native poles).
// Beyond 1.5 rad (86 deg) no box is trusted: an infinite margin sends the visit through the per-point global path.
// Checked against the host map on tiles up to the poles: tests/test_host_abi.py::test_car_tile_margin_bounds_the_map.
__host__ __device__ inline double car_tile_margin(double box_c, double th_abs) {
    if (!(th_abs < 1.5)) return __builtin_inf();
    const double c = cos(th_abs);
    return 1.0 + box_c * 2.0 / (c * c);
}

struct CarrDev {
    const double* sin_lon;  // [n_lon] sin(lon')
    const double* cos_lon;  // [n_lon] cos(lon')
    const float* cos_lat;   // [n_lat] float32 cos(lat)
    const float* sin_lat;   // [n_lat] float32 sin(lat)
    int n_lon, n_lat;
    double dist, cb, sb, cr, sr, cdelt1, cdelt2;
};

struct H9 {
    double h[9];
};

// ---- utils/rectify.py:340-363: the lag-independent part of SphericalTransform.forward for grid point (i, j):
// t0 = degrees(atan(x''/z)) * 3600 / cdelt1, t1 likewise; pixel = (X0 + t0, Y0 + t1).  Operation order follows the
// reference (no fused multiply-add) so that float64 results track NumPy's.
__device__ __forceinline__ bool carr_term(const CarrDev& c, int i, int j, double& t0, double& t1) {
#pragma clang fp contract(off)
    const double cl = (double)c.cos_lat[j];
    const double y = (double)c.sin_lat[j];
    const double x = cl * c.sin_lon[i];
    const double z = cl * c.cos_lon[i];
    const double zz = z * c.cb + y * c.sb;
    const double yy = y * c.cb - z * c.sb;
    const bool vis = zz >= 0.0;  // zclip = 0
    const double yr = yy * c.cr - x * c.sr;
    const double xr = x * c.cr + yy * c.sr;
    const double zd = c.dist - zz;
    t0 = atan(xr / zd) * (180.0 / 3.14159265358979323846) * 3600.0 / c.cdelt1;
    t1 = atan(yr / zd) * (180.0 / 3.14159265358979323846) * 3600.0 / c.cdelt2;
    return vis;
}

__device__ __forceinline__ void apply_h(const H9& m, double x, double y, double& ox, double& oy) {
    const double w = fma(m.h[6], x, fma(m.h[7], y, m.h[8]));
    // w = 1 + O(1e-5) (h[8] = 1, small fields of view): hardware reciprocal + one Newton step is accurate to ~1 ulp
    // there, far below the 1e-9 px the map itself is known to; a NaN map stays NaN
    double r = __builtin_amdgcn_rcp(w);
    r = fma(r, fma(-w, r, 1.0), r);
    ox = fma(m.h[0], x, fma(m.h[1], y, m.h[2])) * r;
    oy = fma(m.h[3], x, fma(m.h[4], y, m.h[5])) * r;
}

// h[8] == 1 exactly (host normalisation): w = 1 + eps with eps = h6 x + h7 y
__device__ __forceinline__ void apply_h_series(const H9& m, double x, double y, double& ox, double& oy) {
    const double eps = fma(m.h[6], x, m.h[7] * y);
    const double q = fma(eps, eps, -eps);  // 1/(1 + eps) - 1 up to eps^3
    const double xn = fma(m.h[0], x, fma(m.h[1], y, m.h[2]));
    const double yn = fma(m.h[3], x, fma(m.h[4], y, m.h[5]));
    ox = fma(xn, q, xn);
    oy = fma(yn, q, yn);
}
// wcslib sphx2s / sphs2x + cars2x for one point: native angles of the target -> unit vector -> rotated -> native angles
// of the shifted map -> its pixel
__device__ __forceinline__ void apply_car(const H9& m, const LaunchU& u, double phi, double theta, double& ox, double& oy) {
    double sp, cp, st, ct;
    sincos(phi, &sp, &cp);
    sincos(theta, &st, &ct);
    const double n0 = ct * cp, n1 = ct * sp, n2 = st;
    const double q0 = fma(m.h[0], n0, fma(m.h[1], n1, m.h[2] * n2));
    const double q1 = fma(m.h[3], n0, fma(m.h[4], n1, m.h[5] * n2));
    const double q2 = fma(m.h[6], n0, fma(m.h[7], n1, m.h[8] * n2));
    const double p = atan2(q1, q0);
    const double t = atan2(q2, sqrt(fma(q0, q0, q1 * q1)));
    ox = fma(u.m00, p, fma(u.m01, t, u.b0));
    oy = fma(u.m10, p, fma(u.m11, t, u.b1));
}
// the same from the unit vector (n0, n1, n2) of the target pixel, which k_precompute stores for MODE_CAR (the two sincos
// of apply_car are lag-independent)
__device__ __forceinline__ void apply_car_vec(const H9& m, const LaunchU& u, double n0, double n1, double n2, double& ox,
                                              double& oy) {
    const double q0 = fma(m.h[0], n0, fma(m.h[1], n1, m.h[2] * n2));
    const double q1 = fma(m.h[3], n0, fma(m.h[4], n1, m.h[5] * n2));
    const double q2 = fma(m.h[6], n0, fma(m.h[7], n1, m.h[8] * n2));
    const double p = atan2(q1, q0);
    const double t = atan2(q2, sqrt(fma(q0, q0, q1 * q1)));
    ox = fma(u.m00, p, fma(u.m01, t, u.b0));
    oy = fma(u.m10, p, fma(u.m11, t, u.b1));
}
template <int MODE>
__device__ __forceinline__ void apply_map(const H9& m, const LaunchU& u, double x, double y, double& ox, double& oy) {
    if (MODE == MODE_CAR) apply_car(m, u, x, y, ox, oy);
    else if (MODE == MODE_HOMOGRAPHY_SERIES) apply_h_series(m, x, y, ox, oy);
    else apply_h(m, x, y, ox, oy);
}

// ---- spline weights of scipy's get_spline_interpolation_weights (ni_splines.c), orders 1 and 2 -----------------
template <int ORDER>
struct Spline;
template <>
struct Spline<2> {
    static constexpr int N = 3;
    // first tap index and weights for coordinate c
    static __device__ __forceinline__ void eval(double c, int& start, double w[3]) {
        const double f = floor(c + 0.5);
        const double t = c - f;
        // scipy: w1 = 0.75 - t^2, w0 = 0.5 (0.5 - t)^2, w2 = 1 - w0 - w1 (= 0.5 (0.5 + t)^2 = w0 + t); evaluated here
        // in 5 operations, equal to scipy's values to ~1 ulp
        const double u = t * t;
        w[1] = 0.75 - u;
        w[0] = fma(0.5, u, fma(-0.5, t, 0.125));
        w[2] = w[0] + t;
        start = (int)f - 1;
    }
};
template <int ORDER>
__device__ __forceinline__ void spline_weights_t(double t, double* w);
template <>
__device__ __forceinline__ void spline_weights_t<2>(double t, double* w) {  // t in [-0.5, 0.5)
    const double u = t * t;
    w[1] = 0.75 - u;
    w[0] = fma(0.5, u, fma(-0.5, t, 0.125));
    w[2] = w[0] + t;
}
template <>
__device__ __forceinline__ void spline_weights_t<1>(double t, double* w) {  // t in [0, 1)
    w[0] = 1.0 - t;
    w[1] = t;
}

// cubic B-spline (scipy ni_splines.c, order 3), argument y in [0, 1): SIX times the weights -- 6 w0 = z^3, 6 w1 =
// 3 y^2 (y - 2) + 4, 6 w2 = 3 z^2 (z - 2) + 4 with z = 1 - y, 6 w3 = 6 - the others (scipy: w3 = 1 - w0 - w1 - w2) -- so
// that scipy's four divisions by 6 per axis become ONE multiplication of the sample by 1/36 (values equal to scipy's
// to ~1 ulp; a float64 division costs ~30 instructions here)
__device__ __forceinline__ void spline_weights6_o3(double y, double* w) {
    const double z = 1.0 - y;
    const double y2 = y * y, z2 = z * z;
    w[0] = z2 * z;
    w[1] = fma(3.0 * y2, y - 2.0, 4.0);
    w[2] = fma(3.0 * z2, z - 2.0, 4.0);
    w[3] = ((6.0 - w[0]) - w[1]) - w[2];
}
template <>
__device__ __forceinline__ void spline_weights_t<3>(double y, double* w) {  // y in [0, 1)
    spline_weights6_o3(y, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] *= (1.0 / 6.0);
}
template <>
struct Spline<3> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void eval(double c, int& start, double w[4]) {
        const double f = floor(c);
        spline_weights_t<3>(c - f, w);
        start = (int)f - 1;
    }
};

template <>
struct Spline<1> {
    static constexpr int N = 2;
    static __device__ __forceinline__ void eval(double c, int& start, double w[2]) {
        const double f = floor(c);
        const double t = c - f;
        w[0] = 1.0 - t;
        w[1] = t;
        start = (int)f;
    }
};

__device__ __forceinline__ int mirror_idx(int i, int n) {  // scipy: reflect about the edge sample
    i = i < 0 ? -i : i;
    i = i > n - 1 ? 2 * (n - 1) - i : i;
    return min(max(i, 0), n - 1);
}

// ---- any spline order 0..5 at run time (reprojection_order is a user argument, alignment.py:54): global-memory gather
// only, not tuned.  Kernels take ORDER == ORDER_RT and read the order from their launch uniforms.
constexpr int ORDER_RT = 0;
template <>
struct Spline<ORDER_RT> {
    static constexpr int N = 6;  // array bound only
};
// scipy ni_splines.c get_spline_interpolation_weights + the start index of ni_interpolation.c
__device__ inline void spline_weights_rt(int order, double c, int& start, double w[6]) {
    const bool odd = (order & 1) != 0;
    const double f = floor(odd ? c : c + 0.5);
    const double y = c - f;
    start = (int)f - order / 2;
    switch (order) {
        case 0:
            w[0] = 1.0;
            break;
        case 1:
            w[0] = 1.0 - y;
            w[1] = y;
            break;
        case 2: {
            w[1] = 0.75 - y * y;
            const double t = 0.5 - y;
            w[0] = 0.5 * t * t;
            w[2] = 1.0 - w[0] - w[1];
            break;
        }
        // (orders 3..5: scipy divides by 6, 24, 12, 120; a float64 division costs ~30 instructions on this GPU, so the
        // divisions are multiplications by the rounded reciprocal here: weights equal to scipy's to ~1 ulp)
        case 3: {
            const double z = 1.0 - y;
            w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) * (1.0 / 6.0);
            w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) * (1.0 / 6.0);
            w[0] = z * z * z * (1.0 / 6.0);
            w[3] = 1.0 - w[0] - w[1] - w[2];
            break;
        }
        case 4: {
            double t = y * y;
            w[2] = t * (t * 0.25 - 0.625) + 115.0 / 192.0;
            const double y1 = 1.0 + y;
            w[1] = y1 * (y1 * (y1 * (5.0 - y1) * (1.0 / 6.0) - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
            const double z = 1.0 - y;
            w[3] = z * (z * (z * (5.0 - z) * (1.0 / 6.0) - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
            const double y2 = 0.5 - y;
            t = y2 * y2;
            w[0] = t * t * (1.0 / 24.0);
            w[4] = 1.0 - w[0] - w[1] - w[2] - w[3];
            break;
        }
        default: {  // 5
            double t = y * y;
            w[2] = t * (t * (0.25 - y * (1.0 / 12.0)) - 0.5) + 0.55;
            const double z = 1.0 - y;
            t = z * z;
            w[3] = t * (t * (0.25 - z * (1.0 / 12.0)) - 0.5) + 0.55;
            const double y1 = y + 1.0;
            w[1] = y1 * (y1 * (y1 * (y1 * (y1 * (1.0 / 24.0) - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
            const double z1 = z + 1.0;
            w[4] = z1 * (z1 * (z1 * (z1 * (z1 * (1.0 / 24.0) - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
            t = z * z;
            w[0] = z * t * t * (1.0 / 120.0);
            w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
            break;
        }
    }
}
__device__ __forceinline__ int mirror_far(int i, int n) {  // scipy NI_EXTEND_MIRROR for indices several samples out
    if (n <= 1) return 0;
    const int p = 2 * (n - 1);
    i = i < 0 ? -i : i;
    i = i % p;
    return i > n - 1 ? p - i : i;
}
// taps of the compile-time orders: one reflection is enough up to order 2 (taps at most one sample outside), order 3
// reaches two samples out
template <int ORDER>
__device__ __forceinline__ int mirror_tap(int i, int n) {
    if constexpr (ORDER > 2) return mirror_far(i, n);
    else return mirror_idx(i, n);
}
// Source images may be CROPS of the image the header describes (the once-only reference preparation uploads only the
// rectangle the target grid can touch): `img` then holds columns x0 .. and rows y0 .. of the W x H image with row pitch
// `pitch`; bounds rule and mirroring use the full W x H, the crop is guaranteed to hold every tap of an in-bounds sample.
struct Crop {
    int x0, y0, pitch;  // pitch <= 0: the whole image (pitch = W)
};
template <typename TS>
__device__ inline double spline_global_rt(const TS* __restrict__ img, int W, int H, double nx, double ny, int order,
                                          bool& inb, Crop cr = {0, 0, 0}) {
    inb = (nx >= 0.0) & (nx <= (double)(W - 1)) & (ny >= 0.0) & (ny <= (double)(H - 1));
    const int pitch = cr.pitch > 0 ? cr.pitch : W;
    const double cx = inb ? nx : (double)cr.x0, cy = inb ? ny : (double)cr.y0;
    int sx, sy;
    double wx[6], wy[6];
    spline_weights_rt(order, cx, sx, wx);
    spline_weights_rt(order, cy, sy, wy);
    if (!inb) return 0.0;  // (discarded by every caller; no tap is read)
    double acc = 0.0;
    for (int a = 0; a <= order; ++a) {
        const TS* __restrict__ rowp = img + (size_t)(mirror_far(sy + a, H) - cr.y0) * pitch - cr.x0;
        double row = 0.0;
        for (int b = 0; b <= order; ++b) row = fma((double)rowp[mirror_far(sx + b, W)], wx[b], row);
        acc = fma(row, wy[a], acc);
    }
    return acc;
}

// The same sample from an LDS window whose apron (order/2 + 1 samples below, order - order/2 + 1 above, mirrored by the
// staging loop) covers every tap of an in-bounds coordinate: lds[(gy - oy) * pitch + (gx - ox)] = image(mirror(gy, gx)).
typedef const __attribute__((address_space(3))) double* LdsF64;
// N x N taps, one row at a time: the N reads of a row are issued together (a tap-by-tap loop under a run-time order
// serialises one LDS round trip per tap)
template <int N>
__device__ __forceinline__ double gather_rows_lds(LdsF64 p, int pitch, const double* wx, const double* wy) {
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        double t[N];
#pragma unroll
        for (int b = 0; b < N; ++b) t[b] = p[a * pitch + b];
        double row = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) row = fma(t[b], wx[b], row);
        acc = fma(row, wy[a], acc);
    }
    return acc;
}
__device__ inline double spline_lds_rt(unsigned win, int pitch, int ox, int oy, double nx, double ny, int order) {
    int sx, sy;
    double wx[6], wy[6];
    spline_weights_rt(order, nx, sx, wx);
    spline_weights_rt(order, ny, sy, wy);
#if defined(__HIP_DEVICE_COMPILE__)
    LdsF64 p = (LdsF64)win + ((sy - oy) * pitch + (sx - ox));
#else
    LdsF64 p = (LdsF64)(uintptr_t)win + ((sy - oy) * pitch + (sx - ox));  // (host pass: never executed)
#endif
    switch (order) {  // (uniform)
        case 0: return gather_rows_lds<1>(p, pitch, wx, wy);
        case 1: return gather_rows_lds<2>(p, pitch, wx, wy);
        case 2: return gather_rows_lds<3>(p, pitch, wx, wy);
        case 3: return gather_rows_lds<4>(p, pitch, wx, wy);
        case 4: return gather_rows_lds<5>(p, pitch, wx, wy);
        default: return gather_rows_lds<6>(p, pitch, wx, wy);
    }
}

// One sample of map_coordinates(order, mode='constant', prefilter=False) from global memory.
// inb = the whole-sample bounds rule (c < 0 or c > n-1 or NaN -> cval).
template <int ORDER, typename TS>
__device__ __forceinline__ double spline_global(const TS* __restrict__ img, int W, int H, double nx, double ny,
                                                bool& inb, Crop cr = {0, 0, 0}) {
    constexpr int N = Spline<ORDER>::N;
    inb = (nx >= 0.0) & (nx <= (double)(W - 1)) & (ny >= 0.0) & (ny <= (double)(H - 1));
    const int pitch = cr.pitch > 0 ? cr.pitch : W;
    // (an out-of-bounds sample is discarded by every caller; it gathers from the crop's first pixels)
    const double cx = inb ? nx : (double)(cr.x0 + (cr.pitch > 0)), cy = inb ? ny : (double)(cr.y0 + (cr.pitch > 0));
    int sx, sy;
    double wx[N], wy[N];
    Spline<ORDER>::eval(cx, sx, wx);
    Spline<ORDER>::eval(cy, sy, wy);
    int ix[N], iy[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        ix[k] = mirror_tap<ORDER>(sx + k, W) - cr.x0;
        iy[k] = (mirror_tap<ORDER>(sy + k, H) - cr.y0) * pitch;
    }
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        double row = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) row = fma((double)img[iy[a] + ix[b]], wx[b], row);
        acc = fma(row, wy[a], acc);
    }
    return acc;
}

// Same sample from an LDS window that already holds the mirrored one-pixel apron:
// lds[(gy - oy) * pitch + (gx - ox)] = image(mirror(gy), mirror(gx)).
template <int ORDER, typename TS>
__device__ __forceinline__ double spline_lds(const TS* lds, int pitch, int ox, int oy, int W, int H, double nx,
                                             double ny, bool& inb) {
    constexpr int N = Spline<ORDER>::N;
    inb = (nx >= 0.0) & (nx <= (double)(W - 1)) & (ny >= 0.0) & (ny <= (double)(H - 1));
    const double cx = inb ? nx : (double)(ox + 1), cy = inb ? ny : (double)(oy + 1);
    int sx, sy;
    double wx[N], wy[N];
    Spline<ORDER>::eval(cx, sx, wx);
    Spline<ORDER>::eval(cy, sy, wy);
    const TS* p = lds + (sy - oy) * pitch + (sx - ox);
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        double row = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) row = fma((double)p[a * pitch + b], wx[b], row);
        acc = fma(row, wy[a], acc);
    }
    return acc;
}

// ---- finite-mean (pivot) ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_sum_finite(const T* __restrict__ v, long long n, double* part_sum,
                                                    long long* part_cnt) {
    __shared__ double ss[256];
    __shared__ long long sc[256];
    double s = 0.0;
    long long c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double x = (double)v[i];
        if (isfinite(x)) {
            s += x;
            ++c;
        }
    }
    ss[threadIdx.x] = s;
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            ss[threadIdx.x] += ss[threadIdx.x + o];
            sc[threadIdx.x] += sc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = ss[0];
        part_cnt[blockIdx.x] = sc[0];
    }
}
__global__ void k_mean_final(const double* part_sum, const long long* part_cnt, int n, double* mean_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        long long c = 0;
        for (int i = 0; i < n; ++i) {
            s += part_sum[i];
            c += part_cnt[i];
        }
        mean_out[0] = c > 0 ? s / (double)c : 0.0;
    }
}

// upload helpers: is every finite value of a float64 image exactly representable in float32?  (flag |= 1 when not)
__global__ void __launch_bounds__(256) k_f32_exact(const double* __restrict__ v, long long n, int* flag) {
    int bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double x = v[i];
        if (x == x && (double)(float)x != x) bad = 1;
    }
    if (bad) atomicOr(flag, 1);
}
__global__ void __launch_bounds__(256) k_f64_to_f32(const double* __restrict__ v, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (float)v[i];
}

// ---- pixels as a FITS data unit stores them (alignment.py:299-314 reads them through astropy.io.fits) -----------------
// big-endian; BITPIX 8 = unsigned bytes, 16 / 32 / 64 = two's complement integers, -32 / -64 = IEEE floats.
// BITPIX = -32 without BSCALE / BZERO: the byte swap IS the decode, in place (the float64 cast of alignment.py:314 is
// exact, the sweep takes float32 pixels as they are).
__global__ void __launch_bounds__(256) k_fits_swap32(unsigned int* __restrict__ v, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        v[i] = __builtin_bswap32(v[i]);
}
// everything else: float64(stored) [* bscale + bzero, two roundings as NumPy's `a.astype(float64) * bscale + bzero`]
__global__ void __launch_bounds__(256) k_fits_to_f64(const void* __restrict__ raw, int bitpix, int scaled, double bscale,
                                                     double bzero, long long n, double* __restrict__ out) {
#pragma clang fp contract(off)  // multiply, round, add, round -- as NumPy does; an FMA would differ in the last bit
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double x;
        switch (bitpix) {
            case 8: x = (double)((const unsigned char*)raw)[i]; break;
            case 16: x = (double)(short)__builtin_bswap16(((const unsigned short*)raw)[i]); break;
            case 32: x = (double)(int)__builtin_bswap32(((const unsigned int*)raw)[i]); break;
            case 64: x = (double)(long long)__builtin_bswap64(((const unsigned long long*)raw)[i]); break;
            case -32: x = (double)__uint_as_float(__builtin_bswap32(((const unsigned int*)raw)[i])); break;
            default: x = __longlong_as_double((long long)__builtin_bswap64(((const unsigned long long*)raw)[i])); break;
        }
        if (scaled) x = x * bscale + bzero;
        out[i] = x;
    }
}

// alignment.py:876-887 in place: |v| < vmin or |v| > vmax -> NaN (comparisons with NaN are false, NaN stays NaN)
template <typename T>
__global__ void __launch_bounds__(256) k_threshold(T* __restrict__ v, long long n, int has_min, double vmin, int has_max,
                                                   double vmax) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double x = fabs((double)v[i]);
        if ((has_min && x < vmin) || (has_max && x > vmax)) v[i] = (T)__builtin_nan("");
    }
}

__global__ void k_fill(double* p, long long n, double v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- once-only resample (reference preparation, alignment.py:646-651; single-header resample) ----------------------
struct ResampleArgs {
    const void* img;  // small / large image, TS
    int W, H;
    int gw, gh;  // output grid [gh][gw]
    CarrDev carr;
    double x0, y0;  // Carrington origin (utils/rectify.py:402-404)
    H9 hom;
    void* out;
    int order_rt;  // ORDER == ORDER_RT: the spline order
    Crop crop;     // `img` holds this crop of the W x H image
    double* bbox;  // k_resample_bbox: [gridDim.x][4] partial (min x, max x, min y, max y) of the in-bounds coordinates
    LaunchU car_fwd, car_inv;  // MODE_CAR: 0-based grid pixel -> its native (phi, theta) [rad]; native angles of the
                               // source map -> its 0-based pixel; `hom` = rotation between the two native frames
};
// the 0-based source-pixel coordinate of output grid point idx -- ONE function for the resample and for the bounding box
// that decides which pixels it can touch (bit-identical coordinates in both)
template <int MODE>
__device__ __forceinline__ bool resample_coord(const ResampleArgs& a, long long idx, double& nx, double& ny) {
    const int i = (int)(idx % a.gw), j = (int)(idx / a.gw);
    bool ok = true;
    if (MODE == MODE_TRANSLATE) {
        double t0, t1;
        ok = carr_term(a.carr, i, j, t0, t1);
        nx = a.x0 + t0;
        ny = a.y0 + t1;
    } else if (MODE == MODE_CAR) {
        const double phi = fma(a.car_fwd.m00, (double)i, fma(a.car_fwd.m01, (double)j, a.car_fwd.b0));
        const double theta = fma(a.car_fwd.m10, (double)i, fma(a.car_fwd.m11, (double)j, a.car_fwd.b1));
        apply_car(a.hom, a.car_inv, phi, theta, nx, ny);
    } else {
        apply_h(a.hom, (double)i, (double)j, nx, ny);
    }
    if (!ok) nx = __builtin_nan("");
    return (nx >= 0.0) & (nx <= (double)(a.W - 1)) & (ny >= 0.0) & (ny <= (double)(a.H - 1));
}
template <int MODE>
__global__ void __launch_bounds__(256) k_resample_bbox(const ResampleArgs a) {
    __shared__ double red[4][4];
    const double inf = __builtin_inf();
    double mnx = inf, mxx = -inf, mny = inf, mxy = -inf;
    const long long n = (long long)a.gw * a.gh;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        double nx, ny;
        if (resample_coord<MODE>(a, idx, nx, ny)) {
            mnx = fmin(mnx, nx);
            mxx = fmax(mxx, nx);
            mny = fmin(mny, ny);
            mxy = fmax(mxy, ny);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fmin(mnx, __shfl_xor(mnx, o));
        mxx = fmax(mxx, __shfl_xor(mxx, o));
        mny = fmin(mny, __shfl_xor(mny, o));
        mxy = fmax(mxy, __shfl_xor(mxy, o));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = mnx;
        red[wave][1] = mxx;
        red[wave][2] = mny;
        red[wave][3] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = a.bbox + (size_t)blockIdx.x * 4;
        o[0] = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
        o[1] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
        o[2] = fmin(fmin(red[0][2], red[1][2]), fmin(red[2][2], red[3][2]));
        o[3] = fmax(fmax(red[0][3], red[1][3]), fmax(red[2][3], red[3][3]));
    }
}
template <int MODE, int ORDER, typename TS, typename TO>
__global__ void __launch_bounds__(256) k_resample(const ResampleArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.gw * a.gh) return;
    double nx, ny;
    bool inb = resample_coord<MODE>(a, idx, nx, ny);
    double v = __builtin_nan("");
    if (inb) {  // (an out-of-bounds point reads nothing: the source may be a crop that only holds what can be touched)
        if constexpr (ORDER == ORDER_RT) v = spline_global_rt<TS>((const TS*)a.img, a.W, a.H, nx, ny, a.order_rt, inb, a.crop);
        else v = spline_global<ORDER, TS>((const TS*)a.img, a.W, a.H, nx, ny, inb, a.crop);
    }
    ((TO*)a.out)[idx] = (TO)v;
}

// ---- precompute: base coordinates, culling, tile-major compaction -----------------------------------------------
// Prologue of a sweep, folded into its FIRST k_precompute launch: the lag parameters and output indices the host has
// just written to page-locked memory are read over PCIe by the kernel (a few tens of KB; two DMA-engine copies in their
// place cost ~50 us of queue switches each between the kernels of a sweep) and the output is NaN-initialised (quirk Q9).
struct PrologueArgs {
    const double* src;      // pinned host memory (device-visible): [n_params doubles][n_outidx int64]; null: nothing to do
    double* dst_params;
    long long n_params;
    long long* dst_outidx;
    long long n_outidx;
    double* out;
    long long n_out;
    long long* refine_count;  // the sweep's counter of re-evaluated lag-points (k_finalize), reset here
};
__device__ __forceinline__ void run_prologue(const PrologueArgs& p) {
    if (!p.src) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.refine_count) p.refine_count[0] = p.refine_count[1] = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = i0; i < p.n_params; i += stride) p.dst_params[i] = p.src[i];
    const long long* __restrict__ src_idx = (const long long*)(p.src + p.n_params);
    for (long long i = i0; i < p.n_outidx; i += stride) p.dst_outidx[i] = src_idx[i];
    const double nan = __builtin_nan("");
    for (long long i = i0; i < p.n_out; i += stride) p.out[i] = nan;
}

struct PrecomputeArgs {
    PrologueArgs prologue;  // first launch of a sweep only
    const void* ref;  // reference on grid, TA, [gh][gw]
    int gw, gh;
    int tile_w, tile_h;  // tile_w * tile_h == kTilePts
    int tiles_x, tiles_y;
    CarrDev carr;             // MODE_TRANSLATE
    LaunchU car_fwd;             // MODE_CAR: 0-based target pixel -> its native (phi, theta) [rad]
    double f0lo, f0hi, f1lo, f1hi;  // cull box on the base coordinates (inclusive)
    int residus;              // 1: method 'residus' -> pts hold the raw reference value and 1/sqrt(value)
    int tile_skip;            // MODE_TRANSLATE: 1 = drop whole tiles that provably miss the cull box
    double lip_x, lip_y;      // pixels per radian of grid-point motion (upper bounds), see k_precompute
    double dlon, dlat;        // grid steps in radians (upper bounds)
    const double* pivot_a;    // device scalar: mean of the finite reference values
    Pt* pts;                  // [n_tiles][kTilePts] compacted points
    int* tile_count;          // [n_tiles]
    double* tile_bbox;        // [n_tiles][4] min0, max0, min1, max1 over the kept points
};
template <int MODE, typename TA>
__global__ void __launch_bounds__(256) k_precompute(const PrecomputeArgs a) {
    __shared__ int wave_cnt[4];
    __shared__ double red[4][4];
    run_prologue(a.prologue);
    const int tile = blockIdx.x;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double pivot = a.pivot_a[0];
    const double inf = __builtin_inf();
    if (MODE == MODE_TRANSLATE && a.tile_skip) {
        // Whole tile outside the cull box?  A bound, not a heuristic: (t0, t1) = K atan(x''/zd), K atan(y''/zd) with
        // (x'', y'', zz) components of a rotated unit vector and zd = dist - zz >= dist - 1 > 0, so moving the grid
        // point by an angle d (radians, <= |d lon| + |d lat|) moves x''/zd by at most d (1/(dist-1) + 1/(dist-1)^2) and
        // atan is 1-Lipschitz: |t - t_centre| <= lip * (|di| dlon + |dj| dlat) for every point of the tile, visible or
        // not (host: lip_x/lip_y, dlon/dlat with their rounding margins; tile_skip = 0 when dist <= 1).
        __shared__ int s_skip;
        const int i0 = tx * a.tile_w, j0 = ty * a.tile_h;
        const int i1 = min(i0 + a.tile_w, a.gw) - 1, j1 = min(j0 + a.tile_h, a.gh) - 1;
        if (threadIdx.x == 0) {
            const int ic = (i0 + i1) / 2, jc = (j0 + j1) / 2;
            double t0, t1;
            carr_term(a.carr, ic, jc, t0, t1);
            const double ang = (double)max(ic - i0, i1 - ic) * a.dlon + (double)max(jc - j0, j1 - jc) * a.dlat;
            const double m0 = a.lip_x * ang + 1.0, m1 = a.lip_y * ang + 1.0;
            // (a NaN centre compares false everywhere: no skip)
            s_skip = (t0 + m0 < a.f0lo) || (t0 - m0 > a.f0hi) || (t1 + m1 < a.f1lo) || (t1 - m1 > a.f1hi);
        }
        __syncthreads();
        if (s_skip) {
            if (threadIdx.x == 0) a.tile_count[tile] = 0;
            return;
        }
    }
    double mn0 = inf, mx0 = -inf, mn1 = inf, mx1 = -inf;
    int base_pos = 0;
    const size_t tbase = (size_t)tile * kTilePts;
    for (int r = 0; r < kTilePts / 256; ++r) {
        const int k = r * 256 + threadIdx.x;
        const int gi = tx * a.tile_w + (k % a.tile_w);
        const int gj = ty * a.tile_h + (k / a.tile_w);
        bool valid = (gi < a.gw) & (gj < a.gh);
        double b0 = 0.0, b1 = 0.0, av = 0.0;
        if (valid) {
            av = (double)((const TA*)a.ref)[(size_t)gj * a.gw + gi];
            if (MODE == MODE_TRANSLATE) {
                valid = carr_term(a.carr, gi, gj, b0, b1);
            } else if (MODE == MODE_CAR) {  // native (phi, theta) [rad] of the target pixel
                b0 = fma(a.car_fwd.m00, (double)gi, fma(a.car_fwd.m01, (double)gj, a.car_fwd.b0));
                b1 = fma(a.car_fwd.m10, (double)gi, fma(a.car_fwd.m11, (double)gj, a.car_fwd.b1));
            } else {
                b0 = (double)gi;
                b1 = (double)gj;
            }
            valid = valid & isfinite(av) & (b0 >= a.f0lo) & (b0 <= a.f0hi) & (b1 >= a.f1lo) & (b1 <= a.f1hi);
        }
        const unsigned long long bal = __ballot(valid);
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_pos;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (valid) {
            Pt pt;
            pt.b0 = b0;
            pt.b1 = b1;
            pt.a = a.residus ? av : av - pivot;
            pt.pad = a.residus ? 1.0 / sqrt(av) : 0.0;  // alignment.py:545 norm = sqrt(data_large)
            if (MODE == MODE_CAR) {
                // unit vector of the pixel (the lag-independent half of apply_car); the bounding box below stays in
                // (phi, theta); method 'residus' recomputes 1/sqrt(a) in the sweep
                double sp, cp, st, ct;
                sincos(b0, &sp, &cp);
                sincos(b1, &st, &ct);
                pt.b0 = ct * cp;
                pt.b1 = ct * sp;
                pt.pad = st;
            }
            a.pts[tbase + off + rank] = pt;
            mn0 = fmin(mn0, b0);
            mx0 = fmax(mx0, b0);
            mn1 = fmin(mn1, b1);
            mx1 = fmax(mx1, b1);
        }
        base_pos += tot;
        __syncthreads();
    }
    if (MODE != MODE_CAR && !a.residus) {
        // Sums of (reference - pivot) and of its square over every full chunk of kChunk compacted points, in a fixed
        // order: where k_sweep knows that every sample of a visit is finite (interior window without a NaN) the count
        // and these two moments do not depend on the lag, and the lanes add them per chunk instead of per sample.
        // (The stores above are visible: the loop ends with a workgroup barrier.)
        // Round 6: the sums live in the pads of the chunk's SECOND and THIRD point; the pad of its FIRST point says
        // whether the chunk is a RUN -- kChunk neighbouring pixels of one grid row, (x, y), (x + 1, y), ... -- which lets
        // the homography sweeps advance their affine terms by one addition each instead of re-evaluating them
        // (tile_points, kIncr).  The first point's pad is what the rolling scalar prefetch holds when the chunk starts.
        static_assert(kChunk >= 3, "run flag + the two chunk sums live in the pads of the chunk's first three points");
        Pt* tp = a.pts + tbase;
        for (int c = threadIdx.x; c < base_pos / kChunk; c += 256) {
            double sa = tp[c * kChunk].a, saa = sa * sa;
            bool run = MODE == MODE_HOMOGRAPHY || MODE == MODE_HOMOGRAPHY_SERIES;
#pragma unroll
            for (int k = 1; k < kChunk; ++k) {
                const double v = tp[c * kChunk + k].a;
                sa += v;
                saa = fma(v, v, saa);
                run = run && tp[c * kChunk + k].b0 == tp[c * kChunk].b0 + (double)k && tp[c * kChunk + k].b1 == tp[c * kChunk].b1;
            }
            tp[c * kChunk].pad = run ? 1.0 : 0.0;
            tp[c * kChunk + 1].pad = sa;
            tp[c * kChunk + 2].pad = saa;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn0 = fmin(mn0, __shfl_xor(mn0, o));
        mx0 = fmax(mx0, __shfl_xor(mx0, o));
        mn1 = fmin(mn1, __shfl_xor(mn1, o));
        mx1 = fmax(mx1, __shfl_xor(mx1, o));
    }
    if (lane == 0) {
        red[wave][0] = mn0;
        red[wave][1] = mx0;
        red[wave][2] = mn1;
        red[wave][3] = mx1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.tile_count[tile] = base_pos;
        double* bb = a.tile_bbox + (size_t)tile * 4;
        bb[0] = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
        bb[1] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
        bb[2] = fmin(fmin(red[0][2], red[1][2]), fmin(red[2][2], red[3][2]));
        bb[3] = fmax(fmax(red[0][3], red[1][3]), fmax(red[2][3], red[3][3]));
    }
}

// Work partition (single workgroup, deterministic):
//   tile_list  = non-empty tiles in tile order; tile_cum[k] = work units before list entry k, one unit =
//   kChunk * kPointGroups points (tile_cum[n_nonempty] = total);  group_first[g] = list entry in which the units of
//   tile group g start when the total is cut in n_groups equal shares;
//   info[0] = non-empty tiles, info[1] = kept points, info[2] = work units;  info[3..6] = 0: k_sweep adds its tile
//   visits there (all, LDS-staged, interior, all-finite interior -- coreg_last_visit_counts).
constexpr int kUnitPts = kChunk * kPointGroups;
// Where tile group g's share of the `total` work units starts.  taper_frac = 0: equal shares.  Otherwise the last
// taper_frac / 1024 of the groups get linearly smaller shares, down to taper_min / 1024 of a full one, and the others
// proportionally more: the groups are dispatched in increasing order, so small late workgroups even out the end of a
// launch of many rounds, and the larger early ones stage fewer partial tiles per point (measured).
// Evaluated once per group by k_tile_list, which leaves the starts in a table for k_sweep.
__host__ __device__ inline long long group_start(int g, long long total, int n_groups, int taper_min, int taper_frac) {
    if (taper_frac <= 0) return (long long)g * total / n_groups;
    const int g0 = n_groups - (int)((long long)n_groups * taper_frac / 1024);  // first tapered group
    const int nt = n_groups - 1 - g0;                                          // its weight falls over nt steps
    if (nt <= 0) return (long long)g * total / n_groups;
    // weight(k) = 1 for k < g0, 1 - (1 - m) (k - g0) / nt after: cumulative weight in closed form (exact in float64)
    const double m = (double)taper_min / 1024.0;
    auto cum = [&](int k) -> double {
        if (k <= g0) return (double)k;
        const double j = (double)(k - g0);
        return (double)g0 + j - (1.0 - m) * j * (j - 1.0) / (2.0 * (double)nt);
    };
    if (g >= n_groups) return total;
    const long long s0 = (long long)(cum(g) / cum(n_groups) * (double)total);
    return s0 < 0 ? 0 : (s0 > total ? total : s0);
}
__global__ void __launch_bounds__(1024) k_tile_list(const int* __restrict__ tile_count, int n_tiles, int n_groups,
                                                    int* tile_list, int* tile_cum, int* group_first, long long* info,
                                                    int taper_min, int taper_frac) {
    __shared__ int wcnt[16];
    __shared__ int wunits[16];
    __shared__ long long wpts[16];
    __shared__ int s_base, s_units;
    __shared__ long long s_pts;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        s_base = 0;
        s_units = 0;
        s_pts = 0;
    }
    __syncthreads();
    for (int t0 = 0; t0 < n_tiles; t0 += 1024) {
        const int t = t0 + threadIdx.x;
        const int c = t < n_tiles ? tile_count[t] : 0;
        const bool nz = c > 0;
        const int units = (c + kUnitPts - 1) / kUnitPts;
        const unsigned long long bal = __ballot(nz);
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        long long pts = c;
        int uincl = units;  // inclusive wave scan of the units
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(uincl, o);
            if (lane >= o) uincl += v;
        }
        for (int o = 32; o > 0; o >>= 1) pts += __shfl_xor(pts, o);
        if (lane == 63) wunits[wave] = uincl;
        if (lane == 0) {
            wcnt[wave] = __popcll(bal);
            wpts[wave] = pts;
        }
        __syncthreads();
        int off = s_base, uoff = s_units;
        for (int w = 0; w < wave; ++w) {
            off += wcnt[w];
            uoff += wunits[w];
        }
        if (nz) {
            tile_list[off + rank] = t;
            tile_cum[off + rank] = uoff + uincl - units;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0, tu = 0;
            long long tp = 0;
            for (int w = 0; w < 16; ++w) {
                tot += wcnt[w];
                tu += wunits[w];
                tp += wpts[w];
            }
            s_base += tot;
            s_units += tu;
            s_pts += tp;
        }
        __syncthreads();
    }
    const int n_list = s_base, total = s_units;
    if (threadIdx.x == 0) {
        tile_cum[n_list] = total;
        info[0] = n_list;
        info[1] = s_pts;
        info[2] = total;
        info[3] = info[4] = info[5] = info[6] = 0;
    }
    __syncthreads();
    // first list entry of each group's unit range [g * total / n_groups, ...): largest k with tile_cum[k] <= start
    for (int g = threadIdx.x; g <= n_groups; g += 1024) {
        const long long start = group_start(g, total, n_groups, taper_min, taper_frac);
        group_first[1024 + g] = (int)start;  // second half of the table: the unit each group's share starts at
        int lo = 0, hi = n_list;  // answer in [0, n_list]
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (mid <= n_list && (long long)tile_cum[mid] <= start) lo = mid;
            else hi = mid - 1;
        }
        group_first[g] = lo;
    }
}

// ---- the sweep -------------------------------------------------------------------------------------------------------
struct SweepArgs {
    const void* img;  // small image, TS [H][W]
    int W, H;
    const Pt* pts;  // tile-major compacted points
    const int* tile_count;
    const int* tile_list;
    const int* tile_cum;         // work units before each list entry (see k_tile_list)
    const int* group_first;      // first list entry of each tile group
    long long* tile_info;  // [0] = non-empty tiles, [1] = kept points, [2] = work units, [3..6] visit counters
    const double* tile_bbox;
    const double* lane_params;  // SoA [NP][n_slots]; TRANSLATE: X0, Y0; HOMOGRAPHY: h0..h8
    long long n_slots;          // n_batches * 256
    int n_batches;
    int n_groups;  // multiple of 8: the compacted points are cut in n_groups equal shares
    int group_lo;  // first group this launch sweeps (multi-GPU point sharding: a rank's share of the groups; else 0)
    double* partials;  // [groups of this launch][kNumSums][n_slots]
    const double* pivots;  // device: [0] mean(reference) (already subtracted in aval), [1] mean(small image)
    int use_lds;
    int clean_path;    // 1: interior visits whose window holds only finite values skip the sample mask (point_lag, CLEAN)
    int lds_elems;     // capacity of the dynamic LDS window in float64 elements
    LaunchU car_inv;      // MODE_CAR: native (phi, theta) [rad] -> 0-based pixel of the shifted map of this launch
};

struct Acc {
    int n;
    double a, b, aa, bb, ab;
};

// The N x N float64 taps of one sample, from LDS, as N*N separate ds_read_b64.  hipcc would fuse neighbouring 8-byte
// reads into ds_read2_b64, which occupies the LDS for 8 cycles against 2 + 2 for two ds_read_b64
// (MI355X_MICROARCH.md, LDS table), so the reads are issued by hand; wait() / wait_after() are the matching s_waitcnt
// and tie the values to it so that no use can be scheduled above the wait.
template <int N>
struct Taps;
template <>
struct Taps<3> {
    double t[9];
    __device__ __forceinline__ void issue(unsigned a0, unsigned a1, unsigned a2) {
        asm volatile(
            "ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:8\n\tds_read_b64 %2, %9 offset:16\n\t"
            "ds_read_b64 %3, %10\n\tds_read_b64 %4, %10 offset:8\n\tds_read_b64 %5, %10 offset:16\n\t"
            "ds_read_b64 %6, %11\n\tds_read_b64 %7, %11 offset:8\n\tds_read_b64 %8, %11 offset:16"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
              "=&v"(t[8])
            : "v"(a0), "v"(a1), "v"(a2));
    }
    // the wait, placed after everything (w0..w5) depends on has been computed
    __device__ __forceinline__ void wait_after(double& w0, double& w1, double& w2, double& w3, double& w4, double& w5) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                       "+v"(t[8]), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5));
    }
    // same, and pins (fx, fy) to the issue point: arithmetic that starts from them (the spline weights) is scheduled
    // AFTER the reads have been issued and overlaps their latency
    __device__ __forceinline__ void issue_before(unsigned a0, unsigned a1, unsigned a2, double& fx, double& fy) {
        asm volatile(
            "ds_read_b64 %0, %11\n\tds_read_b64 %1, %11 offset:8\n\tds_read_b64 %2, %11 offset:16\n\t"
            "ds_read_b64 %3, %12\n\tds_read_b64 %4, %12 offset:8\n\tds_read_b64 %5, %12 offset:16\n\t"
            "ds_read_b64 %6, %13\n\tds_read_b64 %7, %13 offset:8\n\tds_read_b64 %8, %13 offset:16"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
              "=&v"(t[8]), "+v"(fx), "+v"(fy)
            : "v"(a0), "v"(a1), "v"(a2));
    }
    __device__ __forceinline__ void wait() {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                       "+v"(t[8]));
    }
    // compile-time window pitch P (elements): the three rows are immediate offsets of ONE address register, which
    // saves the two row-address additions per sample
    template <int P>
    __device__ __forceinline__ void issue_before_imm(unsigned a0, double& fx, double& fy) {
        static_assert(P > 0 && (2 * P + 2) * 8 < 65536, "LDS immediate offsets are 16 bits");
        asm volatile(
            "ds_read_b64 %0, %11\n\tds_read_b64 %1, %11 offset:8\n\tds_read_b64 %2, %11 offset:16\n\t"
            "ds_read_b64 %3, %11 offset:%12\n\tds_read_b64 %4, %11 offset:%13\n\tds_read_b64 %5, %11 offset:%14\n\t"
            "ds_read_b64 %6, %11 offset:%15\n\tds_read_b64 %7, %11 offset:%16\n\tds_read_b64 %8, %11 offset:%17"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
              "=&v"(t[8]), "+v"(fx), "+v"(fy)
            : "v"(a0), "n"(P * 8), "n"(P * 8 + 8), "n"(P * 8 + 16), "n"(2 * P * 8), "n"(2 * P * 8 + 8),
              "n"(2 * P * 8 + 16));
    }
};
template <>
struct Taps<2> {
    double t[4];
    __device__ __forceinline__ void issue(unsigned a0, unsigned a1, unsigned) {
        asm volatile(
            "ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\t"
            "ds_read_b64 %2, %5\n\tds_read_b64 %3, %5 offset:8"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
            : "v"(a0), "v"(a1));
    }
    // (only instantiated, never reached: the bilinear path calls issue() / wait())
    __device__ __forceinline__ void issue_before(unsigned a0, unsigned a1, unsigned a2, double&, double&) {
        issue(a0, a1, a2);
    }
    __device__ __forceinline__ void wait_after(double&, double&, double&, double&, double&, double&) { wait(); }
    __device__ __forceinline__ void wait() {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
    }
};

template <>
struct Taps<4> {
    double t[16];
    // the sixteen reads of a cubic sample; (fx, fy) pinned to the issue point so that the weight arithmetic that starts
    // from them is scheduled after the reads and overlaps their latency
    __device__ __forceinline__ void issue_before(unsigned a0, unsigned a1, unsigned a2, unsigned a3, double& fx,
                                                 double& fy) {
        asm volatile(
            "ds_read_b64 %0, %18\n\tds_read_b64 %1, %18 offset:8\n\tds_read_b64 %2, %18 offset:16\n\t"
            "ds_read_b64 %3, %18 offset:24\n\t"
            "ds_read_b64 %4, %19\n\tds_read_b64 %5, %19 offset:8\n\tds_read_b64 %6, %19 offset:16\n\t"
            "ds_read_b64 %7, %19 offset:24\n\t"
            "ds_read_b64 %8, %20\n\tds_read_b64 %9, %20 offset:8\n\tds_read_b64 %10, %20 offset:16\n\t"
            "ds_read_b64 %11, %20 offset:24\n\t"
            "ds_read_b64 %12, %21\n\tds_read_b64 %13, %21 offset:8\n\tds_read_b64 %14, %21 offset:16\n\t"
            "ds_read_b64 %15, %21 offset:24"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
              "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]),
              "=&v"(t[15]), "+v"(fx), "+v"(fy)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    }
    // compile-time window pitch P: the four rows are immediate offsets of one address register
    template <int P>
    __device__ __forceinline__ void issue_before_imm(unsigned a0, double& fx, double& fy) {
        static_assert(P > 0 && (3 * P + 3) * 8 < 65536, "LDS immediate offsets are 16 bits");
        asm volatile(
            "ds_read_b64 %0, %18\n\tds_read_b64 %1, %18 offset:8\n\tds_read_b64 %2, %18 offset:16\n\t"
            "ds_read_b64 %3, %18 offset:24\n\t"
            "ds_read_b64 %4, %18 offset:%19\n\tds_read_b64 %5, %18 offset:%20\n\tds_read_b64 %6, %18 offset:%21\n\t"
            "ds_read_b64 %7, %18 offset:%22\n\t"
            "ds_read_b64 %8, %18 offset:%23\n\tds_read_b64 %9, %18 offset:%24\n\tds_read_b64 %10, %18 offset:%25\n\t"
            "ds_read_b64 %11, %18 offset:%26\n\t"
            "ds_read_b64 %12, %18 offset:%27\n\tds_read_b64 %13, %18 offset:%28\n\tds_read_b64 %14, %18 offset:%29\n\t"
            "ds_read_b64 %15, %18 offset:%30"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
              "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]),
              "=&v"(t[15]), "+v"(fx), "+v"(fy)
            : "v"(a0), "n"(P * 8), "n"(P * 8 + 8), "n"(P * 8 + 16), "n"(P * 8 + 24), "n"(2 * P * 8), "n"(2 * P * 8 + 8),
              "n"(2 * P * 8 + 16), "n"(2 * P * 8 + 24), "n"(3 * P * 8), "n"(3 * P * 8 + 8), "n"(3 * P * 8 + 16),
              "n"(3 * P * 8 + 24));
    }
    __device__ __forceinline__ void wait_after(double* wx, double* wy) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                       "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]),
                       "+v"(t[15]), "+v"(wx[0]), "+v"(wx[1]), "+v"(wx[2]), "+v"(wx[3]), "+v"(wy[0]), "+v"(wy[1]),
                       "+v"(wy[2]), "+v"(wy[3]));
    }
};

// Cubic B-spline sample (reprojection_order = 3, alignment.py:54) from the LDS window at the window-relative coordinate
// (ux, uy) = coordinate - 1 - window origin: trunc(u) is the window index of the first of the four taps per axis,
// fract(u) the spline argument.  Sixteen hand-issued reads, the weights (six-fold, see spline_weights6_o3) computed under
// their latency, one multiplication by 1/36 at the end.
template <int PITCH, bool FOLDED = false>
__device__ __forceinline__ double gather_o3(unsigned win, int pitch, double cx, double cy, double offx, double offy) {
    // floor and fraction of the COORDINATE itself, the (integer) window offset added afterwards: near the left / top edge
    // of the image the cubic apron puts the window origin at -2, the offset is +1, and c + 1 can round a coordinate one
    // ulp below an integer up to it -- another first tap than scipy's floor(c) (met by whole-pixel lags, round 5)
    // (FOLDED: the coordinate handed over is window-relative already and non-negative -- the Carrington sweep's interior
    // path, whose lane origin carries the offset: truncation and v_fract, two instructions fewer per axis)
    int c0, r0;
    double fx, fy;
    if constexpr (FOLDED) {
        c0 = (int)cx;
        r0 = (int)cy;
        fx = __builtin_amdgcn_fract(cx);
        fy = __builtin_amdgcn_fract(cy);
    } else {
        const double flx = floor(cx), fly = floor(cy);
        c0 = (int)flx + (int)offx;
        r0 = (int)fly + (int)offy;
        fx = cx - flx;
        fy = cy - fly;
    }
    const unsigned a0 = win + 8u * (unsigned)(__mul24(r0, PITCH > 0 ? PITCH : pitch) + c0);
    Taps<4> tp;
    if constexpr (PITCH > 0) {
        tp.template issue_before_imm<PITCH>(a0, fx, fy);
    } else {
        const unsigned a1 = a0 + 8u * (unsigned)pitch;
        const unsigned a2 = a1 + 8u * (unsigned)pitch;
        const unsigned a3 = a2 + 8u * (unsigned)pitch;
        tp.issue_before(a0, a1, a2, a3, fx, fy);
    }
    double wx[4], wy[4];
    spline_weights6_o3(fx, wx);
    spline_weights6_o3(fy, wy);
    tp.wait_after(wx, wy);
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double row = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) row = fma(tp.t[r * 4 + c], wx[c], row);
        v = fma(row, wy[r], v);
    }
    return v * (1.0 / 36.0);
}

// Quadratic B-spline sample from the LDS window at the window-relative coordinate (ux, uy) = coordinate + 1/2 - 1 -
// window origin: trunc(u) is the window index of the first tap, f = fract(u) = t + 1/2 the spline argument.  The weights
// of scipy's quadratic B-spline, w0 = (1/2 - t)^2 / 2, w1 = 3/4 - t^2, w2 = (1/2 + t)^2 / 2, are g^2 / 2, 1/2 + f g,
// f^2 / 2 with g = 1 - f; they are evaluated DOUBLED (5 operations per axis instead of 6) and the window of an order-2
// visit holds the pixels times 1/4 (exact), which puts the factor 2 x 2 back.  The nine reads are issued before the
// weight arithmetic and waited for after it.
template <int PITCH>
__device__ __forceinline__ double gather_o2(unsigned win, int pitch, double ux, double uy) {
    const int c0 = (int)ux, r0 = (int)uy;
    const unsigned a0 = win + 8u * (unsigned)(__mul24(r0, PITCH > 0 ? PITCH : pitch) + c0);
    Taps<3> tp;
    double fx = __builtin_amdgcn_fract(ux), fy = __builtin_amdgcn_fract(uy);
    if constexpr (PITCH > 0) {
        tp.template issue_before_imm<PITCH>(a0, fx, fy);
    } else {
        const unsigned a1 = a0 + 8u * (unsigned)pitch;
        const unsigned a2 = a1 + 8u * (unsigned)pitch;
        tp.issue_before(a0, a1, a2, fx, fy);
    }
    double wx[3], wy[3];
    const double gx = 1.0 - fx, gy = 1.0 - fy;
    wx[0] = gx * gx;
    wx[2] = fx * fx;
    wx[1] = (2.0 - wx[0]) - wx[2];
    wy[0] = gy * gy;
    wy[2] = fy * fy;
    wy[1] = (2.0 - wy[0]) - wy[2];
    tp.wait_after(wx[0], wx[1], wx[2], wy[0], wy[1], wy[2]);
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double row = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) row = fma(tp.t[r * 3 + c], wx[c], row);
        v = fma(row, wy[r], v);
    }
    return v;
}

// One (grid point, lag) of alignment.py:519-531 for this lane's lag.  Everything is under the lane's own
// predicate (EXEC mask): lanes whose coordinate violates the bounds rule, or whose sample is not finite,
// simply skip -- no selects, no clamped addresses.
// LDS path: the float64 window (mirrored apron included, pivot already subtracted unless ROUND) lives at LDS byte
// address `win`; element (r, c) is at r*pitch + c.
// INTERIOR: the caller has proved that every (point of the tile) x (lag of the workgroup) is inside the image, so the
// bounds rule cannot trigger and is not evaluated (padding lanes never get here).
// RESID (method 'residus', alignment.py:544-547): d = (a - b) / sqrt(a) summed instead of the Pearson moments;
// `isa` = 1/sqrt(a).  No NaN mask exists in that method: k_finalize returns NaN unless EVERY grid point contributed.
// pxw, pyw (interior LDS visits): TRANSLATE: lane origin + (0.5 for order 2) - (first tap's offset + window origin), so
// that trunc(pxw + b0) is the window column of the first tap and fract() gives the spline argument; other modes: the
// same constant to add to the mapped coordinate.
// CLEAN (interior LDS visits whose window holds no NaN or infinity, Pearson method): every sample is finite, so the mask
// is not evaluated and only the three lag-dependent sums are accumulated here; the count and the two moments of the
// reference are added per chunk by tile_points.
// PREMAPPED (interior LDS visits of the homography modes at order 2, tile_points kIncr): (b0, b1) IS the window-relative
// mapped coordinate already.
template <int MODE, int ORDER, typename TS, bool LDS, bool ROUND, bool RESID, bool INTERIOR = false, int PITCH = 0,
          bool CLEAN = false, bool PREMAPPED = false>
__device__ __forceinline__ void point_lag(Acc& acc, unsigned win, const TS* __restrict__ img, int pitch,
                                          int ox, int oy, int W, int H, double wmax, double hmax, double px0, double py0,
                                          double pxw, double pyw, const H9& hm, const LaunchU& cu, double b0, double b1, double av,
                                          double isa, double pivot_b) {
    constexpr int N = Spline<ORDER>::N;
    if constexpr (INTERIOR && LDS) {
        // Every sample is inside the image: no bounds rule.  Window-relative coordinate u = coordinate + 0.5 - 1 -
        // origin (order 2) / coordinate - origin (order 1): trunc(u) = window index of the first tap, fract(u) -> t.
        // (For TRANSLATE the constant is folded into the lane's origin, which moves the float64 rounding of the sum by
        // at most one ulp of the coordinate, ~2e-13 px.)
        double ux, uy, mx = 0.0, my = 0.0;
        if constexpr (PREMAPPED) {
            static_assert(ORDER == 2, "the cubic gather wants the unshifted coordinate");
            ux = b0;
            uy = b1;
        } else if (MODE == MODE_TRANSLATE) {
            ux = pxw + b0;
            uy = pyw + b1;
        } else {
            if (MODE == MODE_CAR) apply_car_vec(hm, cu, b0, b1, isa, mx, my);  // (b0, b1, isa) = unit vector
            else apply_map<MODE>(hm, cu, b0, b1, mx, my);
            ux = mx + pxw;
            uy = my + pyw;
        }
        double v = 0.0;
        if constexpr (ORDER == 2) {
            v = gather_o2<PITCH>(win, pitch, ux, uy);
        } else if constexpr (ORDER == 3) {
            // (TRANSLATE: the window offset is folded into the lane's origin, the coordinate IS window-relative)
            if (MODE == MODE_TRANSLATE) v = gather_o3<PITCH, true>(win, pitch, ux, uy, 0.0, 0.0);
            else v = gather_o3<PITCH>(win, pitch, mx, my, pxw, pyw);
        } else {
            const int c0 = (int)ux, r0 = (int)uy;
            const unsigned a0 = win + 8u * (unsigned)(__mul24(r0, pitch) + c0);
            const unsigned a1 = a0 + 8u * (unsigned)pitch;
            const unsigned a2 = a1 + 8u * (unsigned)pitch;
            Taps<N> tp;
            double wx[N], wy[N];
            tp.issue(a0, a1, a2);
            spline_weights_t<ORDER>(__builtin_amdgcn_fract(ux), wx);
            spline_weights_t<ORDER>(__builtin_amdgcn_fract(uy), wy);
            tp.wait();
#pragma unroll
            for (int r = 0; r < N; ++r) {
                double row = 0.0;
#pragma unroll
                for (int c = 0; c < N; ++c) row = fma(tp.t[r * N + c], wx[c], row);
                v = fma(row, wy[r], v);
            }
        }
        double bm;
        if (ROUND) {
            v = (double)(float)v;  // float32 dst of alignment.py:1024
            bm = v - pivot_b;
        } else {
            bm = v;  // the window holds (pixel - pivot)
        }
        if constexpr (CLEAN) {
            static_assert(!RESID && MODE != MODE_CAR, "no room for the chunk sums in Pt");
            acc.b += bm;
            acc.bb = fma(bm, bm, acc.bb);
            acc.ab = fma(av, bm, acc.ab);
        } else if (RESID) {
            const double d = (av - (ROUND ? v : v + pivot_b)) * (MODE == MODE_CAR ? 1.0 / sqrt(av) : isa);
            if (isfinite(d)) {
                acc.n += 1;
                acc.b += d;
                acc.bb = fma(d, d, acc.bb);
            }
        } else if (isfinite(v)) {
            acc.n += 1;
            acc.a += av;
            acc.b += bm;
            acc.aa = fma(av, av, acc.aa);
            acc.bb = fma(bm, bm, acc.bb);
            acc.ab = fma(av, bm, acc.ab);
        }
        return;
    }
    double nx, ny;
    if (MODE == MODE_TRANSLATE) {
        nx = px0 + b0;  // self.x + term, utils/rectify.py:362
        ny = py0 + b1;
    } else {
        if (MODE == MODE_CAR) apply_car_vec(hm, cu, b0, b1, isa, nx, ny);  // (b0, b1, isa) = unit vector
        else apply_map<MODE>(hm, cu, b0, b1, nx, ny);
    }
    if (INTERIOR || ((nx >= 0.0) & (nx <= wmax) & (ny >= 0.0) & (ny <= hmax))) {
        int sx, sy;
        double wx[N], wy[N];
        double v = 0.0;
        if constexpr (ORDER == ORDER_RT) {
            if constexpr (LDS) {
                v = spline_lds_rt(win, pitch, ox, oy, nx, ny, cu.order_rt);
            } else {
                bool inb;
                v = spline_global_rt<TS>(img, W, H, nx, ny, cu.order_rt, inb);
            }
        } else if constexpr (LDS && ORDER == 2) {
            // the arithmetic of the interior path under this lane's bounds predicate: (pxw, pyw) = 1/2 - 1 - window
            // origin turn the coordinate into the window-relative one
            v = gather_o2<PITCH>(win, pitch, nx + pxw, ny + pyw);
        } else if constexpr (LDS && ORDER == 3) {
            v = gather_o3<PITCH>(win, pitch, nx, ny, pxw, pyw);  // (pxw, pyw) = -1 - window origin
        } else if constexpr (LDS) {
            // tap addresses first, so that the reads are in flight while the weights are computed
            const double fx = floor(nx + (ORDER == 2 ? 0.5 : 0.0)), fy = floor(ny + (ORDER == 2 ? 0.5 : 0.0));
            const int r0 = (int)fy - (ORDER == 2 ? 1 : 0) - oy;
            const int c0 = (int)fx - (ORDER == 2 ? 1 : 0) - ox;
            // byte addresses of the first tap of each row (window rows and pitch are far below 2^23)
            const unsigned a0 = win + 8u * (unsigned)(__mul24(r0, pitch) + c0);
            const unsigned a1 = a0 + 8u * (unsigned)pitch;
            const unsigned a2 = a1 + 8u * (unsigned)pitch;
            Taps<N> tp;
            tp.issue(a0, a1, a2);
            Spline<ORDER>::eval(nx, sx, wx);
            Spline<ORDER>::eval(ny, sy, wy);
            tp.wait();
#pragma unroll
            for (int r = 0; r < N; ++r) {
                double row = 0.0;
#pragma unroll
                for (int c = 0; c < N; ++c) row = fma(tp.t[r * N + c], wx[c], row);
                v = fma(row, wy[r], v);
            }
        } else {
            Spline<ORDER>::eval(nx, sx, wx);
            Spline<ORDER>::eval(ny, sy, wy);
            int ix[N], iy[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                ix[k] = mirror_tap<ORDER>(sx + k, W);
                iy[k] = mirror_tap<ORDER>(sy + k, H) * W;
            }
#pragma unroll
            for (int r = 0; r < N; ++r) {
                double row = 0.0;
#pragma unroll
                for (int c = 0; c < N; ++c) row = fma((double)img[iy[r] + ix[c]], wx[c], row);
                v = fma(row, wy[r], v);
            }
        }
        double bm;
        if (ROUND) {
            v = (double)(float)v;  // float32 dst of alignment.py:1024
            bm = v - pivot_b;
        } else {
            // LDS window already holds (pixel - pivot): the spline weights sum to 1
            bm = LDS ? v : v - pivot_b;
        }
        if (RESID) {
            const double braw = (ROUND || !LDS) ? v : v + pivot_b;  // undo the pivot folded into the LDS window
            const double d = (av - braw) * (MODE == MODE_CAR ? 1.0 / sqrt(av) : isa);
            if (isfinite(d)) {
                acc.n += 1;
                acc.b += d;
                acc.bb = fma(d, d, acc.bb);
            }
        } else if (isfinite(v)) {
            acc.n += 1;
            acc.a += av;
            acc.b += bm;
            acc.aa = fma(av, av, acc.aa);
            acc.bb = fma(bm, bm, acc.bb);
            acc.ab = fma(av, bm, acc.ab);
        }
    }
}

// One compacted point through the scalar data path.  The address is wave-uniform (tile, chunk and point-group are), but
// the compiler cannot prove the buffer read-only (k_precompute wrote it, another kernel) and would emit per-lane vector
// loads of one address: two VMEM instructions per point that occupy the texture addresser for 64 lanes' worth of
// cycles.  Read through the constant address space the same bytes come as s_load_dwordx4/x2 into SGPRs (the scalar
// cache is invalidated at kernel start, so it sees k_precompute's stores) and the VALU takes them as scalar operands.
#ifndef COREG_PT_SCALAR
#define COREG_PT_SCALAR 2
#endif
#ifndef COREG_PT_ANCHOR
#define COREG_PT_ANCHOR 1
#endif
// COREG_PT_SCALAR == 2: explicit s_load_dwordx8 of a whole Pt, issued a point ahead (tile_points)
typedef unsigned SPt __attribute__((ext_vector_type(8)));
// `anchor`: a per-lane value the point's FIRST vector instruction reads; naming it as an in/out operand keeps the load
// above that instruction (the scheduler would otherwise sink it below the coordinate arithmetic and shorten the time
// the load has before the gather's wait)
__device__ __forceinline__ SPt spt_load(const Pt* __restrict__ p, double& anchor) {
    SPt r;
#if defined(__HIP_DEVICE_COMPILE__)
#if COREG_PT_ANCHOR
    asm volatile("s_load_dwordx8 %0, %2, 0x0" : "=s"(r), "+v"(anchor) : "s"(p));
#else
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(r) : "s"(p));
#endif
#else
    r = (SPt)(0u);
    (void)p;
    (void)anchor;
#endif
    return r;
}
__device__ __forceinline__ void spt_wait(SPt& r) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
#endif
}
__device__ __forceinline__ double spt_f64(const SPt& r, int i) {
    const unsigned long long u = ((unsigned long long)r[2 * i + 1] << 32) | (unsigned long long)r[2 * i];
    return __builtin_bit_cast(double, u);
}
__device__ __forceinline__ void load_pt_uniform(const Pt* __restrict__ p, Pt& out) {
#if defined(__HIP_DEVICE_COMPILE__) && COREG_PT_SCALAR
    typedef const __attribute__((address_space(4))) double* ConstF64;
    ConstF64 q = (ConstF64)(const double*)p;
    out.b0 = q[0];
    out.b1 = q[1];
    out.a = q[2];
    out.pad = q[3];
#else
    out = *p;
#endif
}

// Walk this point-group's share of the compacted points of one tile: chunks of kChunk points, chunk c belongs to
// point-group (c % kPointGroups).  Point data are wave-uniform: the loads below use uniform addresses (scalar loads).
template <int MODE, int ORDER, typename TS, bool LDS, bool ROUND, bool RESID, bool INTERIOR = false, int PITCH = 0,
          bool CLEAN = false>
__device__ __forceinline__ void tile_points(Acc& acc, unsigned win, const TS* __restrict__ img, int pitch,
                                            int ox, int oy, int W, int H, double px0, double py0, double pxw,
                                            double pyw, const H9& hm, const LaunchU& cu, const Pt* __restrict__ pts, int p_begin, int p_end,
                                            double pivot_b, int pg) {
    // points [p_begin, p_end) of the tile: p_begin is a multiple of kChunk * kPointGroups, p_end is one too or the
    // tile's point count
    const double wmax = (double)(W - 1), hmax = (double)(H - 1);
    const int n_full = p_end / kChunk;
#if defined(__HIP_DEVICE_COMPILE__) && COREG_PT_SCALAR == 2
    // Round 6: homography modes, order 2, interior LDS visits.  The projective map of this lane,
    // (xn, yn, w)(x, y), is affine in the grid pixel; two things are taken out of the per-sample arithmetic:
    //  * the window offset (pxw, pyw) is folded into the map ONCE per visit: T(+off) H has rows r0 + pxw r2, r1 + pyw r2,
    //    so the sample's window-relative coordinate is xn' / w directly (6 fma per visit instead of 2 add per sample);
    //  * in a RUN (k_precompute: kChunk neighbouring pixels of one grid row) xn', yn' and w (or eps) of the points after
    //    the first are the previous ones plus the x-column of the map: 3 additions instead of 6 fma.  Every chunk is
    //    seeded afresh, so at most kChunk - 1 additions accumulate: <= 3 ulp of a window-relative coordinate (<= 160),
    //    1e-13 px.  The bounds rule is not involved (interior visits keep 1e-9 px clear of it) and an even order has no
    //    noise-decided tap choice inside the image, so the samples the fix kernels re-evaluate -- all on
    //    non-interior visits at this order -- still see apply_map's own coordinate.  Odd orders keep the exact path.
    constexpr bool kIncr = INTERIOR && LDS && ORDER == 2 && !RESID &&
                           (MODE == MODE_HOMOGRAPHY || MODE == MODE_HOMOGRAPHY_SERIES);
    if constexpr (kIncr) {
        if (cu.h_incr) {
            H9 hv = hm;
            hv.h[0] = fma(pxw, hm.h[6], hm.h[0]);
            hv.h[1] = fma(pxw, hm.h[7], hm.h[1]);
            hv.h[2] = MODE == MODE_HOMOGRAPHY_SERIES ? hm.h[2] + pxw : fma(pxw, hm.h[8], hm.h[2]);
            hv.h[3] = fma(pyw, hm.h[6], hm.h[3]);
            hv.h[4] = fma(pyw, hm.h[7], hm.h[4]);
            hv.h[5] = MODE == MODE_HOMOGRAPHY_SERIES ? hm.h[5] + pyw : fma(pyw, hm.h[8], hm.h[5]);
            int c = p_begin / kChunk + pg;
            if (c < n_full) {
                double& anchor = hv.h[7];
                SPt cur = spt_load(pts + c * kChunk, anchor);
                spt_wait(cur);
                for (; c < n_full; c += kPointGroups) {
                    const Pt* __restrict__ q = pts + c * kChunk;
                    const bool run = cur[7] != 0u;  // (pad of the chunk's first point = 1.0 / 0.0: uniform, SGPR test)
                    double xn = 0.0, yn = 0.0, ww = 0.0;  // ww: eps (series) or w
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) {
                        SPt nxt = spt_load(k + 1 < kChunk ? q + k + 1 : q + kPointGroups * kChunk, anchor);
                        const double x = spt_f64(cur, 0), y = spt_f64(cur, 1);
                        if (k == 0 || !run) {
                            xn = fma(hv.h[0], x, fma(hv.h[1], y, hv.h[2]));
                            yn = fma(hv.h[3], x, fma(hv.h[4], y, hv.h[5]));
                            ww = MODE == MODE_HOMOGRAPHY_SERIES ? fma(hv.h[6], x, hv.h[7] * y)
                                                                : fma(hv.h[6], x, fma(hv.h[7], y, hv.h[8]));
                        } else {
                            xn += hv.h[0];
                            yn += hv.h[3];
                            ww += hv.h[6];
                        }
                        double ux, uy;
                        if (MODE == MODE_HOMOGRAPHY_SERIES) {
                            const double qq = fma(ww, ww, -ww);  // 1 / (1 + eps) - 1 up to eps^3
                            ux = fma(xn, qq, xn);
                            uy = fma(yn, qq, yn);
                        } else {
                            double r = __builtin_amdgcn_rcp(ww);
                            r = fma(r, fma(-ww, r, 1.0), r);
                            ux = xn * r;
                            uy = yn * r;
                        }
                        point_lag<MODE, ORDER, TS, LDS, ROUND, RESID, INTERIOR, PITCH, CLEAN, true>(
                            acc, win, img, pitch, ox, oy, W, H, wmax, hmax, px0, py0, pxw, pyw, hv, cu, ux, uy,
                            spt_f64(cur, 2), spt_f64(cur, 3), pivot_b);
                        if constexpr (CLEAN) {
                            if (k == 1) acc.a += spt_f64(cur, 3);
                            if (k == 2) acc.aa += spt_f64(cur, 3);
                            if (k == kChunk - 1) acc.n += kChunk;
                        }
                        spt_wait(nxt);
                        cur = nxt;
                    }
                }
            }
            // ragged tail (< kChunk points): the exact path below
            if (pg == n_full % kPointGroups) {
                for (int p = n_full * kChunk; p < p_end; ++p) {
                    Pt pt;
                    load_pt_uniform(pts + p, pt);
                    point_lag<MODE, ORDER, TS, LDS, ROUND, RESID, INTERIOR, PITCH>(acc, win, img, pitch, ox, oy, W, H, wmax,
                                                                                   hmax, px0, py0, pxw, pyw, hm, cu, pt.b0,
                                                                                   pt.b1, pt.a, pt.pad, pivot_b);
                }
            }
            return;
        }
    }
    {
        // rolling scalar prefetch: the s_load of point m + 1 is issued before point m's address arithmetic and is
        // drained by the s_waitcnt lgkmcnt(0) that ends point m's LDS gather (SMEM and LDS share that counter; a
        // scalar load further ahead would be drained by the same wait, so one point is the useful distance)
        int c = p_begin / kChunk + pg;
        if (c < n_full) {
            // (lane value read first by the point's arithmetic: the window-relative origin / the projective row)
            H9 hml = hm;
            double& anchor = MODE == MODE_TRANSLATE ? (INTERIOR && LDS ? pxw : px0) : hml.h[MODE == MODE_CAR ? 0 : 7];
            SPt cur = spt_load(pts + c * kChunk, anchor);
            spt_wait(cur);
            for (; c < n_full; c += kPointGroups) {
                const Pt* __restrict__ q = pts + c * kChunk;
#pragma unroll
                for (int k = 0; k < kChunk; ++k) {
                    // (past the group's last chunk this reads up to kPointGroups chunks ahead: inside the allocation
                    // -- DevBuf::reserve pads by 25 % + 256 B -- and never used)
                    SPt nxt = spt_load(k + 1 < kChunk ? q + k + 1 : q + kPointGroups * kChunk, anchor);
                    point_lag<MODE, ORDER, TS, LDS, ROUND, RESID, INTERIOR, PITCH, CLEAN>(
                        acc, win, img, pitch, ox, oy, W, H, wmax, hmax, px0, py0, pxw, pyw, hml, cu, spt_f64(cur, 0),
                        spt_f64(cur, 1), spt_f64(cur, 2), spt_f64(cur, 3), pivot_b);
                    if constexpr (CLEAN) {
                        // the chunk's lag-independent sums (k_precompute left them in the pads of its first two points)
                        if (k == 1) acc.a += spt_f64(cur, 3);
                        if (k == 2) acc.aa += spt_f64(cur, 3);
                        if (k == kChunk - 1) acc.n += kChunk;
                    }
                    spt_wait(nxt);  // (already drained by the gather's wait unless no lane sampled)
                    cur = nxt;
                }
            }
        }
    }
#else
    for (int c = p_begin / kChunk + pg; c < n_full; c += kPointGroups) {
        const Pt* __restrict__ q = pts + c * kChunk;
        Pt pt[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) load_pt_uniform(q + k, pt[k]);
#pragma unroll
        for (int k = 0; k < kChunk; ++k)
            point_lag<MODE, ORDER, TS, LDS, ROUND, RESID, INTERIOR, PITCH, CLEAN>(acc, win, img, pitch, ox, oy, W, H, wmax,
                                                                                  hmax, px0, py0, pxw, pyw, hm, cu, pt[k].b0,
                                                                                  pt[k].b1, pt[k].a, pt[k].pad, pivot_b);
        if constexpr (CLEAN) {
            acc.n += kChunk;
            acc.a += pt[1].pad;
            acc.aa += pt[2].pad;
        }
    }
#endif
    // ragged tail (< kChunk points): owned by the point-group next in the rotation
    if (pg == n_full % kPointGroups) {
        for (int p = n_full * kChunk; p < p_end; ++p) {
            Pt pt;
            load_pt_uniform(pts + p, pt);
            point_lag<MODE, ORDER, TS, LDS, ROUND, RESID, INTERIOR, PITCH>(acc, win, img, pitch, ox, oy, W, H, wmax, hmax,
                                                                           px0, py0, pxw, pyw, hm, cu, pt.b0, pt.b1, pt.a,
                                                                           pt.pad, pivot_b);
        }
    }
}

// PITCH > 0: the LDS window has this compile-time row pitch (the host picks one that holds the planned window and whose
// residue mod 32 spreads the lag lattice over the banks best); 0: pitch = window width | 1, chosen per visit
template <int MODE, int ORDER, typename TS, bool ROUND, bool RESID, int PITCH = 0>
__global__ void __launch_bounds__(kSweepThreads) k_sweep(const SweepArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* lds = (double*)lds_raw;
    constexpr int kWaves = kSweepThreads / 64;
    // bounding boxes of the waves, double-buffered by visit parity: a visit then needs two workgroup barriers, not three
    // (a wave can only write slot k again after the barrier of the visit in between, which every wave reaches after
    // it has read slot k)
    // (only the first kBlock / 64 waves write: the kPointGroups copies of each lag are identical)
    constexpr int kLagWaves = kBlock / 64;
    __shared__ double wred2[2][kLagWaves][4];
    // "this wave staged a NaN or an infinity", per visit parity like the boxes (written before the barrier that ends the
    // staging, read after it; the next write of the same slot lies two barriers later)
    __shared__ int wdirty[2][kWaves];
    // all-finite interior visits take the variant without the sample mask (point_lag, CLEAN); a float64 image rounded
    // to float32 (ROUND) could overflow to infinity in the rounding, so it keeps the mask
    constexpr bool kCleanPath = MODE != MODE_CAR && !RESID && ORDER != ORDER_RT && !(ROUND && sizeof(TS) == 8);
    int visit = 0;
    int n_vis = 0, n_vis_lds = 0, n_vis_int = 0, n_vis_clean = 0;  // (uniform: tile visits of this workgroup by kind)

    // XCD-aware block -> (group, batch): blocks with equal blockIdx % 8 share an XCD (round-robin dispatch), so all
    // lag batches of one tile group land on one XCD and re-use its tiles / image window from that XCD's L2.
    const int b = blockIdx.x;
    const int slot8 = b & 7;
    const int q = b >> 3;
    const int batch = q % a.n_batches;
    const int group = a.group_lo + (q / a.n_batches) * 8 + slot8;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = __builtin_amdgcn_readfirstlane(threadIdx.x / kBlock);  // point-group of this wave (uniform)
    const long long slot = (long long)batch * kBlock + (threadIdx.x % kBlock);

    // this lane's lag
    double px0 = 0.0, py0 = 0.0;
    H9 hm;
#pragma unroll
    for (int k = 0; k < 9; ++k) hm.h[k] = 0.0;
    if (MODE == MODE_TRANSLATE) {
        px0 = a.lane_params[slot];
        py0 = a.lane_params[a.n_slots + slot];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) hm.h[k] = a.lane_params[(long long)k * a.n_slots + slot];
    }

    // padding lanes carry NaN parameters (never in bounds)
    const bool pad_lane = MODE == MODE_TRANSLATE ? (px0 != px0) : (hm.h[8] != hm.h[8]);

    Acc acc = {0, 0.0, 0.0, 0.0, 0.0, 0.0};

    const TS* __restrict__ img = (const TS*)a.img;
    const int W = a.W, H = a.H;
    const double inf = __builtin_inf();
    // this tile group's share of the work: units [u_lo, u_hi) of the concatenated compacted points
    const int u_lo = a.group_first[1024 + group];  // (k_tile_list: group_start)
    const int u_hi = a.group_first[1024 + group + 1];
    const int n_list = (int)a.tile_info[0];
    const double pivot_b = a.pivots[1];
    // LDS byte address of the window (through the LDS address space: the generic pointer's null check would otherwise
    // be re-evaluated with every sample's address)
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned win = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw;
#else
    const unsigned win = 0;
#endif

    for (int tl = a.group_first[group]; tl < n_list && u_lo < u_hi; ++tl) {
        const int ubase = a.tile_cum[tl];
        if (ubase >= u_hi) break;
        const int tile = a.tile_list[tl];
        const int cnt = a.tile_count[tile];
        const int p_begin = max(u_lo - ubase, 0) * kUnitPts;
        const int p_end = min((u_hi - ubase) * kUnitPts, cnt);
        if (p_begin >= p_end) continue;
        const double* bb = a.tile_bbox + (size_t)tile * 4;
        const double bx0 = bb[0], bx1 = bb[1], by0 = bb[2], by1 = bb[3];

        // bounding box, in small-image pixels, of (tile points) x (this workgroup's lags)
        double mnx, mxx, mny, mxy;
        if (MODE == MODE_TRANSLATE) {
            mnx = bx0 + px0;
            mxx = bx1 + px0;
            mny = by0 + py0;
            mxy = by1 + py0;
        } else {
            mnx = inf; mxx = -inf; mny = inf; mxy = -inf;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double cx, cy;
                apply_map<MODE>(hm, a.car_inv, (c & 1) ? bx1 : bx0, (c & 2) ? by1 : by0, cx, cy);
                mnx = fmin(mnx, cx);
                mxx = fmax(mxx, cx);
                mny = fmin(mny, cy);
                mxy = fmax(mxy, cy);
            }
            if (MODE == MODE_CAR) {
                // a rotation of the sphere followed by (atan2, asin) is not projective: the images of the corners do
                // not bound the tile's image; car_tile_margin widens the box by the curvature over THIS tile (by0, by1 =
                // its native latitude range; polar tiles get an infinite box = the per-point global path).  A tile that
                // straddles the +-pi cut of atan2 maps its corners to both ends of the map: such a box is never
                // "interior" and does not fit the LDS, so the visit takes the per-point global path.
                const double margin = car_tile_margin(a.car_inv.box_c, fmax(fabs(by0), fabs(by1)) + a.car_inv.pole_sep);
                mnx -= margin;
                mxx += margin;
                mny -= margin;
                mxy += margin;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            mnx = fmin(mnx, __shfl_xor(mnx, o));
            mxx = fmax(mxx, __shfl_xor(mxx, o));
            mny = fmin(mny, __shfl_xor(mny, o));
            mxy = fmax(mxy, __shfl_xor(mxy, o));
        }
        double(*wred)[4] = wred2[visit & 1];
        int* wdirt = wdirty[visit & 1];
        ++visit;
        if (lane == 0 && wave < kLagWaves) {
            wred[wave][0] = mnx;
            wred[wave][1] = mxx;
            wred[wave][2] = mny;
            wred[wave][3] = mxy;
        }
        __syncthreads();  // every wave has left the previous tile's LDS window; the boxes of this visit are in place
        // the kPointGroups copies of each lag are identical: the first kBlock/64 waves cover every lag
        mnx = fmin(fmin(wred[0][0], wred[1][0]), fmin(wred[2][0], wred[3][0]));
        mxx = fmax(fmax(wred[0][1], wred[1][1]), fmax(wred[2][1], wred[3][1]));
        mny = fmin(fmin(wred[0][2], wred[1][2]), fmin(wred[2][2], wred[3][2]));
        mxy = fmax(fmax(wred[0][3], wred[1][3]), fmax(wred[2][3], wred[3][3]));
        // no in-bounds sample possible for this (tile, batch)?  (uniform)
        if (!(mxx >= 0.0) || !(mnx <= (double)(W - 1)) || !(mxy >= 0.0) || !(mny <= (double)(H - 1))) continue;
        // every (point, lag) of this visit inside the image?  (uniform; the box covers all non-padding lanes.)  The box is
        // made of mapped CORNERS: a pixel between them can come out a few ulp beyond (a whole-pixel lag under an unrotated
        // header puts column 0 at x = -3e-16 between corners at 0.0) -- such a visit must keep the per-sample bounds rule,
        // which is also what k_tap_fix assumes when it takes a noise-decided sample out again: "interior" needs clearance
        constexpr double kClear = 1e-9;
        const bool interior = (mnx >= kClear) & (mxx <= (double)(W - 1) - kClear) & (mny >= kClear) &
                              (mxy <= (double)(H - 1) - kClear);
        // integer window with the mirrored apron: taps of in-bounds samples lie in [floor(c)-1, floor(c)+2] (orders 1, 2);
        // run-time orders: [floor(c) - order/2 - 1, floor(c) + order - order/2 + 1]
        // (kWide: the apron can reach several samples past the image edge -- run-time orders and the cubic kernel)
        constexpr bool kWide = ORDER == ORDER_RT || ORDER > 2;
        const int ord = ORDER == ORDER_RT ? a.car_inv.order_rt : ORDER;
        const int ap_lo = kWide ? ord / 2 + 1 : 1;
        const int ap_hi = kWide ? ord - ord / 2 + 1 : 2;
        const int ox = max((int)floor(fmax(mnx, 0.0)) - ap_lo, kWide ? -ap_lo : -1);
        const int oy = max((int)floor(fmax(mny, 0.0)) - ap_lo, kWide ? -ap_lo : -1);
        const int ex = min((int)floor(fmin(mxx, (double)(W - 1))) + ap_hi, kWide ? W - 1 + ap_hi : W);
        const int ey = min((int)floor(fmin(mxy, (double)(H - 1))) + ap_hi, kWide ? H - 1 + ap_hi : H);
        const int ww = ex - ox + 1, wh = ey - oy + 1;
        // Odd pitch, and not any odd pitch: with lags ~2 px apart, rows r and r + 2 of the window hold neighbouring lag
        // rows, so 2 * pitch must not be close to a multiple of 32 bank pairs.  Measured on the headline sweep with
        // compile-time pitches (ms per step): 113 (17 mod 32) 4.26, 115 (19) 3.90, 117 (21) 3.39, 119 (23) 3.39,
        // 121 (25) 3.34, 123 (27) 3.35 -- the ranking the bank-conflict simulation gives.  The
        // per-visit pitch is therefore moved up to the next odd value whose residue lies in [5, 11] or [21, 27] when
        // the window still fits.
        int pitch = PITCH > 0 ? PITCH : (ww | 1);
        if (PITCH == 0) {
            const int r = pitch & 31;
            const int up = r < 5 ? 5 - r : ((r > 11 && r < 21) ? 21 - r : (r > 27 ? 37 - r : 0));
            if ((long long)(pitch + up) * wh <= (long long)a.lds_elems) pitch += up;
        }
        const long long need = (long long)pitch * wh;
        const bool in_lds = a.use_lds && (need <= (long long)a.lds_elems) && (PITCH == 0 || ww <= PITCH);

        const Pt* __restrict__ pts = a.pts + (size_t)tile * kTilePts;

        bool swept = false;
        ++n_vis;
        {
          if (in_lds) {
            swept = true;
            ++n_vis_lds;
            // Stage the window.  The loads are L2 round trips: kStage rows x kCols column chunks per wave are in flight
            // at a time (one wave would otherwise wait out ~20 dependent load -> store round trips per visit).  Deeper
            // than 8 x 1 measured neutral on the headline (6 x 2, 8 x 2, 12 x 2 = a wave's whole share in one round
            // trip: 3.07 - 3.10 ms all, the last one at the price of SGPR spills).
#ifndef COREG_STAGE_ROWS
#define COREG_STAGE_ROWS 8
#endif
#ifndef COREG_STAGE_COLS
#define COREG_STAGE_COLS 1
#endif
            constexpr int kStage = COREG_STAGE_ROWS, kCols = COREG_STAGE_COLS;
            // the quadratic spline uses doubled weights on both axes (see gather_o2)
            const double scale = ORDER == 2 ? 0.25 : 1.0;
            // (the row index is wave-uniform: with it in an SGPR the row addresses are scalar arithmetic)
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            double poison = 0.0;  // becomes NaN when this lane stages a NaN or an infinity (0 * e)
            for (int r0 = wave_u; r0 < wh; r0 += kWaves * kStage) {
                for (int c0 = 0; c0 < ww; c0 += 64 * kCols) {
                    int gx[kCols];
#pragma unroll
                    for (int j = 0; j < kCols; ++j) {
                        const int c = c0 + 64 * j + lane;
                        // (the apron of the run-time orders can reach several samples past the edge: general reflection)
                        gx[j] = kWide ? mirror_far(ox + min(c, ww - 1), W) : mirror_idx(ox + min(c, ww - 1), W);
                    }
                    TS v[kStage][kCols];
#pragma unroll
                    for (int k = 0; k < kStage; ++k) {
                        const int r = min(r0 + k * kWaves, wh - 1);
                        const int gy = kWide ? mirror_far(oy + r, H) : mirror_idx(oy + r, H);
                        const TS* __restrict__ row = img + (size_t)gy * W;
#pragma unroll
                        for (int j = 0; j < kCols; ++j) v[k][j] = row[gx[j]];
                    }
#pragma unroll
                    for (int j = 0; j < kCols; ++j) {
                        const int c = c0 + 64 * j + lane;
                        if (c < ww) {
#pragma unroll
                            for (int k = 0; k < kStage; ++k) {
                                const int r = r0 + k * kWaves;
                                if (r < wh) {
                                    const double e = (ROUND ? (double)v[k][j] : (double)v[k][j] - pivot_b) * scale;
                                    lds[r * pitch + c] = e;
                                    if (kCleanPath) poison = fma(e, 0.0, poison);
                                }
                            }
                        }
                    }
                }
            }
            if (kCleanPath && interior) {
                const int dirty = __ballot(poison != poison) != 0ull;  // (all lanes vote: outside the lane-0 branch)
                if (lane == 0) wdirt[wave] = dirty;
            }
            __syncthreads();
            bool done = false;
            if constexpr (ORDER != ORDER_RT) {
                if (interior) {
                    done = true;
                    bool clean = false;
                    if constexpr (kCleanPath) clean = a.clean_path && __ballot(wdirt[lane % kWaves] != 0) == 0ull;  // (uniform)
                    ++n_vis_int;
                    n_vis_clean += clean ? 1 : 0;
                    if (!pad_lane) {
                        // window-relative lane constants (exact: a small integer is subtracted)
                        // (first tap = floor(c [+ 1/2 for the even order]) - ORDER / 2)
                        const double offx = (ORDER == 2 ? 0.5 : 0.0) - (double)(ORDER / 2 + ox);
                        const double offy = (ORDER == 2 ? 0.5 : 0.0) - (double)(ORDER / 2 + oy);
                        const double pxw = MODE == MODE_TRANSLATE ? px0 + offx : offx;
                        const double pyw = MODE == MODE_TRANSLATE ? py0 + offy : offy;
                        if (kCleanPath && clean) {
                            tile_points<MODE, ORDER, TS, true, ROUND, RESID, true, PITCH, kCleanPath>(
                                acc, win, img, pitch, ox, oy, W, H, px0, py0, pxw, pyw, hm, a.car_inv, pts, p_begin, p_end,
                                pivot_b, pg);
                        } else {
                            tile_points<MODE, ORDER, TS, true, ROUND, RESID, true, PITCH>(acc, win, img, pitch, ox, oy, W, H,
                                                                                          px0, py0, pxw, pyw, hm, a.car_inv,
                                                                                          pts, p_begin, p_end, pivot_b, pg);
                        }
                    }
                }
            }
            if (!done) {
                // (orders 2, 3: window-relative offsets for gather_o2 / gather_o3, exact small numbers)
                const double offx = ORDER == 2 ? 0.5 - (double)(1 + ox) : (ORDER == 3 ? -(double)(1 + ox) : 0.0);
                const double offy = ORDER == 2 ? 0.5 - (double)(1 + oy) : (ORDER == 3 ? -(double)(1 + oy) : 0.0);
                tile_points<MODE, ORDER, TS, true, ROUND, RESID, false, PITCH>(acc, win, img, pitch, ox, oy, W, H, px0, py0,
                                                                               offx, offy, hm, a.car_inv, pts, p_begin,
                                                                               p_end, pivot_b, pg);
            }
          }
        }
        if (!swept) {
            tile_points<MODE, ORDER, TS, false, ROUND, RESID>(acc, win, img, 0, 0, 0, W, H, px0, py0, 0.0, 0.0, hm, a.car_inv, pts,
                                                              p_begin, p_end, pivot_b, pg);
        }
    }

    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long*)a.tile_info + 3, (unsigned long long)n_vis);
        atomicAdd((unsigned long long*)a.tile_info + 4, (unsigned long long)n_vis_lds);
        atomicAdd((unsigned long long*)a.tile_info + 5, (unsigned long long)n_vis_int);
        atomicAdd((unsigned long long*)a.tile_info + 6, (unsigned long long)n_vis_clean);
    }
    // add the kPointGroups partial sums of every lag in a fixed order (deterministic), one slab per workgroup
    __syncthreads();  // the window is dead: reuse the LDS
    const int ls = threadIdx.x % kBlock;
    if (pg > 0) {
        double* st = lds + ((size_t)(pg - 1) * kNumSums) * kBlock + ls;
        st[0] = (double)acc.n;
        st[kBlock] = acc.a;
        st[2 * kBlock] = acc.b;
        st[3 * kBlock] = acc.aa;
        st[4 * kBlock] = acc.bb;
        st[5 * kBlock] = acc.ab;
    }
    __syncthreads();
    if (pg == 0) {
        double sn = (double)acc.n, sa = acc.a, sb = acc.b, saa = acc.aa, sbb = acc.bb, sab = acc.ab;
        for (int g = 0; g < kPointGroups - 1; ++g) {
            const double* st = lds + ((size_t)g * kNumSums) * kBlock + ls;
            sn += st[0];
            sa += st[kBlock];
            sb += st[2 * kBlock];
            saa += st[3 * kBlock];
            sbb += st[4 * kBlock];
            sab += st[5 * kBlock];
        }
        double* out = a.partials + (size_t)(group - a.group_lo) * kNumSums * a.n_slots + slot;
        out[0] = sn;
        out[a.n_slots] = sa;
        out[2 * a.n_slots] = sb;
        out[3 * a.n_slots] = saa;
        out[4 * a.n_slots] = sbb;
        out[5 * a.n_slots] = sab;
    }
}

// ---- zero-lag border fix (helioprojective, target header == shifted header) ------------------------------------------
// The sweep evaluates that lag-point with the exact identity map, which keeps every border pixel of the grid; the
// reference's pixel -> sky -> pixel round trip through wcslib drops the border pixels whose coordinate comes back a
// hair outside [0, n-1] (geometry.hpp, WcslibTan).  The host lists those pixels; this kernel subtracts their
// contributions from the lag-point's six sums by writing MINUS their totals into an extra partial-sum slab that
// k_finalize adds like any other.  One workgroup, fixed summation order.
struct BorderFixArgs {
    const void* img;  // small image, TS [H][W]
    int W, H;
    const void* ref;  // reference on grid, float32 (ref_f32) or float64, [gh][gw]
    int ref_f32;
    const int* dropped;  // linear grid indices j * gw + i of the pixels to take out
    int n_dropped;
    int gw;
    int order;
    int round_f32;  // 1: sample rounded to float32 before the mask (alignment.py:1024)
    int residus;
    const double* pivots;
    const double* hom;  // lane parameters of the launch, SoA [9][n_slots]: the slot's (snapped, affine) map
    double* slab;  // [kNumSums][n_slots], the extra slab
    long long n_slots, slot;
    // second run, for the re-evaluation of an ill-conditioned lag-point (RefineArgs.fix_slab): the same samples about the
    // lag-point's OWN pivots ([2][n_slots], relative to the global ones), only when the slot is flagged; null otherwise
    const double* slot_pivots;
    const int* only_flagged;
};
// (the slot's own pivots in a fix kernel's second run -- subtracted AFTER the global ones, as refine_item does; zero in
// the first run; false: nothing to do for this slot)
__device__ __forceinline__ bool fix_pivots(const double* slot_pivots, const int* only_flagged, long long n_slots,
                                           long long slot, double& own_a, double& own_b) {
    own_a = own_b = 0.0;
    if (slot_pivots) {
        if (only_flagged && !only_flagged[slot]) return false;
        own_a = slot_pivots[slot];
        own_b = slot_pivots[n_slots + slot];
    }
    return true;
}
template <typename TS>
__global__ void __launch_bounds__(256) k_border_fix(const BorderFixArgs a) {
    __shared__ double red[256];
    const double pivot_a = a.pivots[0], pivot_b = a.pivots[1];
    double own_a, own_b;
    if (!fix_pivots(a.slot_pivots, a.only_flagged, a.n_slots, a.slot, own_a, own_b)) return;
    double hm[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) hm[k] = a.hom[(long long)k * a.n_slots + a.slot];
    double s[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) s[k] = 0.0;
    for (int p = threadIdx.x; p < a.n_dropped; p += 256) {
        const int idx = a.dropped[p];
        const int i = idx % a.gw, j = idx / a.gw;
        const double araw = a.ref_f32 ? (double)((const float*)a.ref)[idx] : ((const double*)a.ref)[idx];
        if (!isfinite(araw)) continue;  // never entered the sums (k_precompute drops it)
        bool inb;
        // the coordinates k_sweep used for this pixel (same fma order as apply_h_series with h6 = h7 = 0)
        const double nx = fma(hm[0], (double)i, fma(hm[1], (double)j, hm[2]));
        const double ny = fma(hm[3], (double)i, fma(hm[4], (double)j, hm[5]));
        double v = spline_global_rt<TS>((const TS*)a.img, a.W, a.H, nx, ny, a.order, inb);
        if (!inb) continue;
        if (a.round_f32) v = (double)(float)v;
        if (a.residus) {
            const double d = (araw - v) * (1.0 / sqrt(araw));
            if (isfinite(d)) {
                s[0] += 1.0;
                s[2] += d;
                s[4] = fma(d, d, s[4]);
            }
        } else if (isfinite(v)) {
            const double av = (araw - pivot_a) - own_a, bm = (v - pivot_b) - own_b;
            s[0] += 1.0;
            s[1] += av;
            s[2] += bm;
            s[3] = fma(av, av, s[3]);
            s[4] = fma(bm, bm, s[4]);
            s[5] = fma(av, bm, s[5]);
        }
    }
    for (int k = 0; k < kNumSums; ++k) {
        red[threadIdx.x] = s[k];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.slab[(size_t)k * a.n_slots + a.slot] = -red[0];
        __syncthreads();
    }
}

// ---- odd spline orders at noise-decided lag-points ---------------------------------------------------------------------
// Odd orders take floor(c) as their first tap (scipy ni_interpolation.c).  Where the map keeps an image axis invariant
// the coordinate along it comes back from wcslib as integer + eps, and the SIGN of eps decides which taps are used --
// hence which neighbour's NaN poisons the sample (geometry.hpp WcslibTan; host: per-pixel flags, bit 0: y' < j on an
// invariant row axis, bit 1: x' < i on an invariant column axis, bit 2: the pixel falls to the bounds rule).  The sweep evaluated those pixels at the exact
// integer; this pass re-decides them: wherever the finiteness of the sample differs between the exact coordinate and
// the coordinate nudged below the integer, the pixel's contribution is added or taken out.  Two stages (per-block
// partial sums, then one block adds them in a fixed order INTO the extra slab that k_border_fix has set).
struct ParityFixArgs {
    const void* img;
    int W, H;
    const void* ref;
    int ref_f32;
    const unsigned char* flags;  // [gh][gw]
    int gw, gh;
    int order;
    int round_f32, residus;
    const double* pivots;
    const double* hom;
    long long n_slots, slot;
    double* partial;  // [gridDim.x][kNumSums]
    double* slab;     // [kNumSums][n_slots]
    int n_partial;
    const double* slot_pivots;  // as BorderFixArgs
    const int* only_flagged;
};
template <typename TS>
__global__ void __launch_bounds__(256) k_parity_fix(const ParityFixArgs a) {
    __shared__ double red[256];
    const double pivot_a = a.pivots[0], pivot_b = a.pivots[1];
    double own_a, own_b;
    if (!fix_pivots(a.slot_pivots, a.only_flagged, a.n_slots, a.slot, own_a, own_b)) return;
    double hm[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) hm[k] = a.hom[(long long)k * a.n_slots + a.slot];
    double s[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) s[k] = 0.0;
    const long long n = (long long)a.gw * a.gh;
    const double nudge = 9.5367431640625e-07;  // 2^-20: below the integer, far above any rounding of the coordinate
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const unsigned f = a.flags[idx];
        if (f == 0 || (f & 4)) continue;  // nothing to re-decide / dropped by the bounds rule (k_border_fix)
        const double araw = a.ref_f32 ? (double)((const float*)a.ref)[idx] : ((const double*)a.ref)[idx];
        if (!isfinite(araw)) continue;
        const int i = (int)(idx % a.gw), j = (int)(idx / a.gw);
        const double nx = fma(hm[0], (double)i, fma(hm[1], (double)j, hm[2]));
        const double ny = fma(hm[3], (double)i, fma(hm[4], (double)j, hm[5]));
        bool inb0, inb1;
        double v0 = spline_global_rt<TS>((const TS*)a.img, a.W, a.H, nx, ny, a.order, inb0);  // what the sweep used
        double v1 = spline_global_rt<TS>((const TS*)a.img, a.W, a.H, (f & 2) ? nx - nudge : nx, (f & 1) ? ny - nudge : ny,
                                         a.order, inb1);                                      // what the reference uses
        if (!inb0 || !inb1) continue;  // border pixels on the bounds rule: k_border_fix
        if (a.round_f32) {
            v0 = (double)(float)v0;
            v1 = (double)(float)v1;
        }
        const bool fin0 = isfinite(v0), fin1 = isfinite(v1);
        if (fin0 == fin1) continue;
        const double sign = fin1 ? 1.0 : -1.0, v = fin1 ? v1 : v0;
        if (a.residus) {
            const double d = (araw - v) * (1.0 / sqrt(araw));
            if (isfinite(d)) {
                s[0] += sign;
                s[2] += sign * d;
                s[4] += sign * d * d;
            }
        } else {
            const double av = (araw - pivot_a) - own_a, bm = (v - pivot_b) - own_b;
            s[0] += sign;
            s[1] += sign * av;
            s[2] += sign * bm;
            s[3] += sign * av * av;
            s[4] += sign * bm * bm;
            s[5] += sign * av * bm;
        }
    }
    for (int k = 0; k < kNumSums; ++k) {
        red[threadIdx.x] = s[k];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.partial[(size_t)blockIdx.x * kNumSums + k] = red[0];
        __syncthreads();
    }
}
__global__ void k_parity_fix_final(const ParityFixArgs a) {
    if (blockIdx.x != 0 || threadIdx.x >= kNumSums) return;
    if (a.slot_pivots && a.only_flagged && !a.only_flagged[a.slot]) return;  // (k_parity_fix left at once too)
    double t = 0.0;
    for (int b = 0; b < a.n_partial; ++b) t += a.partial[(size_t)b * kNumSums + threadIdx.x];
    a.slab[(size_t)threadIdx.x * a.n_slots + a.slot] += t;
}

// ---- odd spline orders: single samples decided by wcslib's rounding noise (the general case) -------------------------
// The two passes above deal with lag-points whose WHOLE grid sits on integers (zero CRVAL lags).  Any other lag can bring
// single coordinates -- or, for a pure CRVAL1 / CRVAL2 lag under an unrotated header, curves of them -- back within
// 1e-9 px of an integer, where the sign of the noise of the reference's wcslib round trip (alignment.py:1038-1069)
// picks the taps of an odd-order spline, hence which neighbour's NaN poisons the sample.  k_tap_scan lists every
// (lag slot, grid pixel) whose mapped coordinate lies within `tol` of an integer (tol far above the noise, far below a
// pixel); the host evaluates wcslib's own chain for exactly those (geometry.hpp WcslibTan) and k_tap_fix replaces their
// contributions: minus the sample at the homography's coordinate (what k_sweep added), plus the sample at wcslib's.
struct TapScanArgs {
    const double* hom;          // lane parameters of the launch, SoA [9][n_slots]
    long long n_slots;
    const unsigned char* skip;  // [n_slots] 1: the slot's whole grid is handled by k_border_fix / k_parity_fix
    const void* ref;            // reference on grid (pixels that never enter the sums are not listed)
    int ref_f32;
    int gw, gh;
    double wmax, hmax, tol;
    unsigned int* count;        // [1] entries wanted (may exceed cap)
    uint2* list;                // [cap] {slot, linear grid index}
    unsigned int cap;
    int rows_per_block;
    int i_lo, i_hi, j_lo, j_hi;  // (inclusive) the cull box of the sweep: no pixel outside it maps into the image
    // Round 5: a near-integer coordinate only matters where it can change the RESULT.  The tap that enters or leaves an
    // odd-order footprint when the coordinate crosses the integer carries a weight of order (1e-9)^order: the value moves
    // by 1e-12 of a pixel difference -- unless that tap is NaN (it then poisons the sample: 0 * NaN) or the coordinate
    // sits ON the bounds rule.  So only samples on the bounds, or with a non-finite pixel in the union of the two
    // footprints ((order + 2)^2 pixels around the nearest pixel, edges mirrored), are listed for wcslib's chain.
    const void* img;
    int img_f32, W, H, order, nan_filter;
    // even spline orders: the taps do not depend on which side of an integer the coordinate falls, the BOUNDS rule does
    // (c < 0 or c > n - 1, Util.py:98-102) -- 1: list only the samples within `tol` of a bound of the image
    int bounds_only;
    // MODE_CAR (two plate-carree maps): grid pixel -> native angles of the target (fwd), native angles of the shifted map
    // -> its pixel (cu); `hom` then holds the sphere rotations.  No segment bound exists for that map: every pixel of a
    // slot that is not skipped is tested (the host skips all slots but those whose lag keeps an image axis invariant)
    LaunchU cu, fwd;
    // Round 5: a thread of k_tap_scan owns (lag slot, rows) and used to test the pixels of every segment it could not
    // dismiss by itself -- for the pure CRVAL1 / CRVAL2 lags of an unrotated header that is whole columns of pixels walked
    // by a handful of lanes.  Such segments are now queued (slot, row, first pixel) and tested by k_tap_scan_segments, one
    // wavefront per segment, one lane per pixel; a full queue falls back to the in-thread test.
    uint4* seg_list;
    unsigned int* seg_count;
    unsigned int seg_cap;
};
template <typename TS>
__device__ __forceinline__ bool tap_union_has_nonfinite(const TS* __restrict__ img, int W, int H, int mx, int my, int hw) {
    for (int dy = -hw; dy <= hw; ++dy) {
        int yy = my + dy;
        yy = yy < 0 ? -yy : (yy > H - 1 ? 2 * (H - 1) - yy : yy);
        yy = min(max(yy, 0), H - 1);
        for (int dx = -hw; dx <= hw; ++dx) {
            int xx = mx + dx;
            xx = xx < 0 ? -xx : (xx > W - 1 ? 2 * (W - 1) - xx : xx);
            xx = min(max(xx, 0), W - 1);
            if (!isfinite((double)img[(size_t)yy * W + xx])) return true;
        }
    }
    return false;
}
// The sharper form for a coordinate that is near an integer k along ONE axis only (the common case: a curve of such
// pixels under a pure single-axis lag).  Along that axis the two candidate footprints share the `order` taps
// k - (order-1)/2 .. k + (order-1)/2 and differ in one end tap, k - hw or k + hw (hw = (order+1)/2); along the other
// axis the taps are fixed (floor(c) - (order-1)/2 .. + order).  The two samples differ in FINITENESS -- the only
// difference that matters -- exactly when the common block is finite and one end line is not while the other is.
template <typename TS>
__device__ __forceinline__ bool tap_end_lines_differ(const TS* __restrict__ img, int W, int H, int k, double c_other,
                                                     int order, bool near_is_x) {
    const int hw = (order + 1) / 2, half = (order - 1) / 2;
    const int o0 = (int)floor(c_other) - half;  // first tap along the other axis (Spline<ORDER>::eval)
    const int n_near = near_is_x ? W : H, n_other = near_is_x ? H : W;
    bool end_lo = true, end_hi = true;  // "every pixel of that end line is finite"
    for (int t = 0; t <= order; ++t) {
        const int po = mirror_far(o0 + t, n_other);
        for (int d = -hw; d <= hw; ++d) {
            const int pn = mirror_far(k + d, n_near);
            const double v = (double)(near_is_x ? img[(size_t)po * W + pn] : img[(size_t)pn * W + po]);
            if (isfinite(v)) continue;
            if (d == -hw) end_lo = false;
            else if (d == hw) end_hi = false;
            else return false;  // a common tap is not finite: NaN whichever way the noise falls
        }
    }
    return end_lo != end_hi;
}
// One thread per (lag slot, grid row).  Along a row the mapped coordinate is x(i) = (a i + b) / (c i + d): the offsets
// x - i and y - j are evaluated at the ends of 64-pixel segments and bounded in between by the chord plus
// max|f''| L^2 / 8 (f'' = 2 c (b c - a d) / (c i + d)^3, bounded over the row); only segments whose bound comes within
// `tol` of an integer are tested pixel by pixel with the sweep's own arithmetic.  In the sub-map semantics the target
// grid IS the image's grid, the offsets are the lag in pixels plus 1e-5 .. 1e-3 px of field distortion, and all but a
// few segments in ten thousand are dismissed by their end points (cfg2: 0.3 ms where the pixel-by-pixel scan took 11).
// one grid pixel of one lag slot: is its sample within `tol` of an integer coordinate, in range, and able to change the
// result?  Then it is listed.
template <int MODE>
__device__ __forceinline__ void tap_scan_pixel(const TapScanArgs& a, const H9& hm, long long slot, int i, int j) {
    const long long idx = (long long)j * a.gw + i;
    const double araw = a.ref_f32 ? (double)((const float*)a.ref)[idx] : ((const double*)a.ref)[idx];
    if (!isfinite(araw)) return;
    double x, y, bx = (double)i, by = (double)j;
    if (MODE == MODE_CAR) {  // (as k_precompute forms the pixel's native angles)
        bx = fma(a.fwd.m00, (double)i, fma(a.fwd.m01, (double)j, a.fwd.b0));
        by = fma(a.fwd.m10, (double)i, fma(a.fwd.m11, (double)j, a.fwd.b1));
    }
    apply_map<MODE>(hm, a.cu, bx, by, x, y);  // the coordinates k_sweep uses
    const int inr = (int)(x >= -a.tol) & (int)(x <= a.wmax + a.tol) & (int)(y >= -a.tol) & (int)(y <= a.hmax + a.tol);
    const int near = (int)(fabs(x - rint(x)) < a.tol) | (int)(fabs(y - rint(y)) < a.tol);
    if (!(inr & near)) return;
    const bool on_bound = fabs(x) < a.tol || fabs(x - a.wmax) < a.tol || fabs(y) < a.tol || fabs(y - a.hmax) < a.tol;
    if (a.bounds_only && !on_bound) return;
    if (a.nan_filter && !a.bounds_only) {
        if (!on_bound) {
            const int mx = (int)rint(x), my = (int)rint(y), hw = (a.order + 1) / 2;
            const bool near_x = fabs(x - rint(x)) < a.tol, near_y = fabs(y - rint(y)) < a.tol;
            bool can_change;
            if (near_x != near_y && a.nan_filter > 1) {
                // one axis only, and the other coordinate clear of its own integers by far more than any noise
                const double co = near_x ? y : x;
                if (fabs(co - rint(co)) < 1e-3) {
                    can_change = a.img_f32 ? tap_union_has_nonfinite((const float*)a.img, a.W, a.H, mx, my, hw)
                                           : tap_union_has_nonfinite((const double*)a.img, a.W, a.H, mx, my, hw);
                } else {
                    can_change = a.img_f32
                        ? tap_end_lines_differ((const float*)a.img, a.W, a.H, near_x ? mx : my, co, a.order, near_x)
                        : tap_end_lines_differ((const double*)a.img, a.W, a.H, near_x ? mx : my, co, a.order, near_x);
                }
            } else {
                can_change = a.img_f32 ? tap_union_has_nonfinite((const float*)a.img, a.W, a.H, mx, my, hw)
                                       : tap_union_has_nonfinite((const double*)a.img, a.W, a.H, mx, my, hw);
            }
            if (!can_change) return;
        }
    }
    const unsigned k = atomicAdd(a.count, 1u);
    if (k < a.cap) a.list[k] = make_uint2((unsigned)slot, (unsigned)idx);
}
// the queued segments: one wavefront each, one lane per pixel
template <int MODE>
__global__ void __launch_bounds__(256) k_tap_scan_segments(const TapScanArgs a) {
    const unsigned n = min(*a.seg_count, a.seg_cap);
    const int lane = threadIdx.x & 63;
    for (unsigned sg = blockIdx.x * 4 + (threadIdx.x >> 6); sg < n; sg += gridDim.x * 4) {
        const uint4 e = a.seg_list[sg];  // (slot, row, first pixel, one past the last pixel)
        const long long slot = e.x;
        const int i = (int)e.z + lane;
        if (i >= (int)e.w) continue;
        H9 hm;
#pragma unroll
        for (int k = 0; k < 9; ++k) hm.h[k] = a.hom[(long long)k * a.n_slots + slot];
        tap_scan_pixel<MODE>(a, hm, slot, i, (int)e.y);
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) k_tap_scan(const TapScanArgs a) {
    constexpr int L = 64;
    const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;
    if (slot >= a.n_slots || a.skip[slot]) return;
    H9 hm;
#pragma unroll
    for (int k = 0; k < 9; ++k) hm.h[k] = a.hom[(long long)k * a.n_slots + slot];
    LaunchU cu = {};
    const int j0 = a.j_lo + blockIdx.y * a.rows_per_block, j1 = min(j0 + a.rows_per_block, a.j_hi + 1);
    const double last = (double)a.i_hi;
    for (int j = j0; j < j1; ++j) {
        const double dj = (double)j;
        if (MODE == MODE_CAR) {
            // no bound on the offsets of a sphere rotation followed by atan2: every segment of the row is queued (or
            // tested here when the queue is full)
            for (int i0 = a.i_lo; i0 <= a.i_hi; i0 += L) {
                const int iend = min(i0 + L, a.i_hi + 1);
                const unsigned q = a.seg_list ? atomicAdd(a.seg_count, 1u) : a.seg_cap;
                if (q < a.seg_cap) {
                    a.seg_list[q] = make_uint4((unsigned)slot, (unsigned)j, (unsigned)i0, (unsigned)iend);
                } else {
                    for (int i = i0; i < iend; ++i) tap_scan_pixel<MODE>(a, hm, slot, i, j);
                }
            }
            continue;
        }
        // bound of |f''| along the row, for x and for y (NaN maps fail every comparison below: nothing is listed)
        const double d = fma(hm.h[7], dj, hm.h[8]);
        const double dmin = fmin(fabs(d), fabs(fma(hm.h[6], last, d)));
        const double inv3 = 1.0 / (dmin * dmin * dmin);
        const double bx = fma(hm.h[1], dj, hm.h[2]), by = fma(hm.h[4], dj, hm.h[5]);
        const double f2x = 2.0 * fabs(hm.h[6]) * fabs(fma(bx, hm.h[6], -hm.h[0] * d)) * inv3;
        const double f2y = 2.0 * fabs(hm.h[6]) * fabs(fma(by, hm.h[6], -hm.h[3] * d)) * inv3;
        // (1.25: rounding of the bound itself; 1e-12: of the end-point coordinates)
        const double bulge_x = 1.25 * f2x * (double)(L * L) / 8.0 + 1e-12 + a.tol;
        const double bulge_y = 1.25 * f2y * (double)(L * L) / 8.0 + 1e-12 + a.tol;
        const bool sane = dmin > 0.5 && bulge_x < 0.25 && bulge_y < 0.25;  // else: every segment is tested
        double x0, y0;
        apply_map<MODE>(hm, cu, (double)a.i_lo, dj, x0, y0);
        {
            // the whole row first, with the same chord + curvature bound over its full length: a generic lag keeps the
            // offsets within 1e-3 px of "lag in pixels" along the row and the row is dismissed by its two end points
            const double lr = (double)(a.i_hi - a.i_lo);
            const double row_bx = 1.25 * f2x * lr * lr / 8.0 + 1e-12 + a.tol, row_by = 1.25 * f2y * lr * lr / 8.0 + 1e-12 + a.tol;
            if (dmin > 0.5 && row_bx < 0.25 && row_by < 0.25) {
                double xe, ye;
                apply_map<MODE>(hm, cu, last, dj, xe, ye);
                const double gx0 = x0 - (double)a.i_lo, gx1 = xe - last, gy0 = y0 - dj, gy1 = ye - dj;
                const double rxlo = fmin(gx0, gx1) - row_bx, rxhi = fmax(gx0, gx1) + row_bx;
                const double rylo = fmin(gy0, gy1) - row_by, ryhi = fmax(gy0, gy1) + row_by;
                if (ceil(rxlo) > rxhi && ceil(rylo) > ryhi) continue;  // no integer offset anywhere along this row
            }
        }
        for (int i0 = a.i_lo; i0 <= a.i_hi; i0 += L) {
            const int i1 = min(i0 + L, a.i_hi);
            double x1, y1;
            apply_map<MODE>(hm, cu, (double)i1, dj, x1, y1);
            const double fx0 = x0 - (double)i0, fx1 = x1 - (double)i1, fy0 = y0 - dj, fy1 = y1 - dj;
            const double xlo = fmin(fx0, fx1) - bulge_x, xhi = fmax(fx0, fx1) + bulge_x;
            const double ylo = fmin(fy0, fy1) - bulge_y, yhi = fmax(fy0, fy1) + bulge_y;
            // an integer inside [lo, hi]?  (negated comparisons: a NaN coordinate tests the segment, whose pixels then
            // fail the range test one by one)
            bool hit = !sane || !(ceil(xlo) > xhi) || !(ceil(ylo) > yhi);
            if (hit && sane && a.bounds_only) {
                // even orders: only a BOUND of the image inside the segment's coordinate range matters
                const double cxlo = fmin(x0, x1) - bulge_x, cxhi = fmax(x0, x1) + bulge_x;
                const double cylo = fmin(y0, y1) - bulge_y, cyhi = fmax(y0, y1) + bulge_y;
                hit = (cxlo <= 0.0 && cxhi >= 0.0) || (cxlo <= a.wmax && cxhi >= a.wmax) || (cylo <= 0.0 && cyhi >= 0.0) ||
                      (cylo <= a.hmax && cyhi >= a.hmax);
            }
            if (hit && sane && a.bounds_only) {
                // Round 6.  Every row has a segment in which the coordinate crosses a bound of the image (the lag moves the
                // image's edge across the grid), for every lag: on a narrow raster with tens of thousands of lag-points
                // (BASELINE configs[3]: 78 141 x 832 rows) that is more segments than the queue holds, and the rest used
                // to be tested pixel by pixel here -- 17.6 ms of a 40 ms sweep.  Along the chord the coordinate is
                // c0 + s (i - i0), true value within `bulge` of it (the bound used above): a pixel can only be within
                // tol of the bound b when |c0 + s (i - i0) - b| <= bulge, i.e. in an interval of 2 bulge / |s| pixels
                // about the crossing.  Those few pixels (one or two per crossing) are tested at once; an axis the lag
                // leaves invariant (|s| ~ 0: the whole segment may sit on the bound) still goes to the queue.
                const int iend_c = (i0 + L > a.i_hi) ? a.i_hi + 1 : i1;
                const double n = (double)(i1 - i0);
                int clo[4], chi[4], n_cand = 0;
                bool narrow = n >= 1.0;
#pragma unroll
                for (int b = 0; b < 4 && narrow; ++b) {
                    const double c0 = b < 2 ? x0 : y0, c1 = b < 2 ? x1 : y1, bulge = b < 2 ? bulge_x : bulge_y;
                    const double bound = (b & 1) ? (b < 2 ? a.wmax : a.hmax) : 0.0;
                    clo[b] = 0;
                    chi[b] = -1;
                    if (fmin(c0, c1) - bulge > bound || fmax(c0, c1) + bulge < bound) continue;  // never near this bound
                    const double sl = (c1 - c0) / n;
                    if (!(fabs(sl) > 1e-3)) {
                        narrow = false;
                        break;
                    }
                    const double t = (bound - c0) / sl, w = bulge / fabs(sl) + 1e-6;
                    const int lo = max(i0, i0 + (int)floor(t - w)), hi = min(iend_c - 1, i0 + (int)ceil(t + w));
                    clo[b] = lo;
                    chi[b] = hi;
                    n_cand += max(hi - lo + 1, 0);
                }
                if (narrow && n_cand <= 8) {
                    for (int b = 0; b < 4; ++b)
                        for (int i = clo[b]; i <= chi[b]; ++i) {
                            bool seen = false;  // (a pixel in the interval of two bounds is listed once)
                            for (int b2 = 0; b2 < b; ++b2) seen = seen || (i >= clo[b2] && i <= chi[b2]);
                            if (!seen) tap_scan_pixel<MODE>(a, hm, slot, i, j);
                        }
                    hit = false;
                }
            }
            if (hit) {
                const int iend = (i0 + L > a.i_hi) ? a.i_hi + 1 : i1;  // (the shared end point belongs to the next segment)
                const unsigned q = a.seg_list ? atomicAdd(a.seg_count, 1u) : a.seg_cap;
                if (q < a.seg_cap) {
                    a.seg_list[q] = make_uint4((unsigned)slot, (unsigned)j, (unsigned)i0, (unsigned)iend);
                } else {
                    for (int i = i0; i < iend; ++i) tap_scan_pixel<MODE>(a, hm, slot, i, j);
                }
            }
            x0 = x1;
            y0 = y1;
        }
    }
}
struct TapFixArgs {
    const void* img;
    int W, H;
    const void* ref;
    int ref_f32, gw;
    int order, round_f32, residus;
    const double* pivots;
    const double* hom;
    long long n_slots;
    const int* seg_slot;        // [n_seg] one workgroup per listed slot (fixed summation order)
    const int* seg_begin;       // [n_seg + 1] its entries, sorted by pixel
    const unsigned int* pixel;  // [n] linear grid index
    const double* xw;           // [n] wcslib's coordinates of that pixel under the slot's shifted header
    const double* yw;
    double* slab;               // [kNumSums][n_slots] the extra slab
    const double* slot_pivots;  // as BorderFixArgs
    const int* only_flagged;
    LaunchU cu, fwd;            // MODE_CAR: as TapScanArgs
};
template <typename TS, int MODE>
__global__ void __launch_bounds__(256) k_tap_fix(const TapFixArgs a) {
    __shared__ double red[256];
    const int seg = blockIdx.x;
    const long long slot = a.seg_slot[seg];
    const double pivot_a = a.pivots[0], pivot_b = a.pivots[1];
    double own_a, own_b;
    if (!fix_pivots(a.slot_pivots, a.only_flagged, a.n_slots, slot, own_a, own_b)) return;
    H9 hm;
#pragma unroll
    for (int k = 0; k < 9; ++k) hm.h[k] = a.hom[(long long)k * a.n_slots + slot];
    double s[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) s[k] = 0.0;
    for (int e = a.seg_begin[seg] + (int)threadIdx.x; e < a.seg_begin[seg + 1]; e += 256) {
        const long long idx = a.pixel[e];
        const double araw = a.ref_f32 ? (double)((const float*)a.ref)[idx] : ((const double*)a.ref)[idx];
        if (!isfinite(araw)) continue;
        const int i = (int)(idx % a.gw), j = (int)(idx / a.gw);
        double nx, ny, bx = (double)i, by = (double)j;
        if (MODE == MODE_CAR) {  // (as k_precompute forms the pixel's native angles; apply_car = sincos + apply_car_vec)
            bx = fma(a.fwd.m00, (double)i, fma(a.fwd.m01, (double)j, a.fwd.b0));
            by = fma(a.fwd.m10, (double)i, fma(a.fwd.m11, (double)j, a.fwd.b1));
        }
        apply_map<MODE>(hm, a.cu, bx, by, nx, ny);
        for (int pass = 0; pass < 2; ++pass) {  // 0: take out what the sweep added; 1: add what the reference samples
            bool inb;
            double v = spline_global_rt<TS>((const TS*)a.img, a.W, a.H, pass ? a.xw[e] : nx, pass ? a.yw[e] : ny, a.order, inb);
            if (!inb) continue;
            const double sign = pass ? 1.0 : -1.0;
            if (a.round_f32) v = (double)(float)v;
            if (a.residus) {
                const double d = (araw - v) * (1.0 / sqrt(araw));
                if (isfinite(d)) {
                    s[0] += sign;
                    s[2] += sign * d;
                    s[4] += sign * d * d;
                }
            } else if (isfinite(v)) {
                const double av = (araw - pivot_a) - own_a, bm = (v - pivot_b) - own_b;
                s[0] += sign;
                s[1] += sign * av;
                s[2] += sign * bm;
                s[3] += sign * av * av;
                s[4] += sign * bm * bm;
                s[5] += sign * av * bm;
            }
        }
    }
    for (int k = 0; k < kNumSums; ++k) {
        red[threadIdx.x] = s[k];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.slab[(size_t)k * a.n_slots + slot] += red[0];
        __syncthreads();
    }
}

// ---- finalize: add the tile-group slabs in a fixed order, Pearson coefficient (c_correlate.py:39-72) -------------
// Ill-conditioned lag-points.  The coefficient comes from six sums taken about two GLOBAL pivots (the images' means); when
// the samples of a lag-point lie far from a pivot compared with their own spread -- a handful of samples, or an overlap
// inside a flat region -- the subtraction  sum xx - (sum x)^2 / n  cancels (relative error eps * sum xx / (n var)).
// k_finalize notices (both quotients are at hand), FLAGS such a lag-point and leaves its own two means -- which the
// one-pass sums do give accurately -- as the lag-point's private pivots.  Round 5: every flagged lag-point is then
// re-evaluated, by kernels of their own (k_refine_list -> k_refine, whose last block finalises), with sums centred on THOSE pivots
// and the corrected two-pass formula  cov = S_ab - S_a S_b / n  (the residual S_a, S_b of an approximate mean cancel to
// first order: the result has the accuracy of c_correlate.py:39-72's means-first evaluation).  One pass over the
// compacted points per flagged lag-point, spread over (lag-point x chunk of the tile list) work items in a fixed order:
// no cap, deterministic, and the same on every GPU of a grid-sharded sweep.
struct RefineArgs {
    int enabled;  // 0: never (method 'residus', launches with noise-decided border pixels)
    double cond;  // sum xx / (n var) above which a lag-point is re-evaluated (kRefineCond; tests lower it)
    int mode, order, small_f32;
    const void* img;  // image to align, float / double [H][W]
    int W, H;
    const Pt* pts;
    const int* tile_list;
    const int* tile_count;
    const long long* tile_info;
    const double* lane_params;  // SoA [2 or 9][n_slots]
    const double* pivots;
    LaunchU car_inv;
    // work space of the re-evaluation (per handle, sized for the launch)
    int* flags;           // [n_slots] 1: flagged by k_finalize
    double* slot_pivots;  // [2][n_slots] the lag-point's own means, relative to the global pivots
    // [kNumSums][n_slots] or null: what a launch's noise-decided samples (k_border_fix / k_parity_fix / k_tap_fix, run a
    // second time about the slot pivots) take out of / put into the re-evaluated sums; added by the last block
    const double* fix_slab;
    int* list;            // [n_slots] flagged slots in slot order (k_refine_list)
    int* head;            // [0] number of flagged slots, [1] chunks per slot (k_refine_list); [3]: ticket of k_refine's
                          // "the block that finishes last writes the coefficients" step, zero between launches
    double* partial;      // [max(kRefineItems, n_slots)][kNumSums] partial sums of the work items
    const long long* out_index;  // (k_refine's last block writes the coefficients)
    long long lag_begin;
    double* out;
    long long* refine_count;
};
constexpr double kRefineCond = 1e5;  // default threshold on sum xx / (n var) (one-pass error below it: < 1e-11)
// k_refine's grid.  An EMPTY launch -- the normal case -- costs the dispatch of its waves (18 us for 2048 blocks, 7 us
// for 512), and the all-flagged headline sweep (3600 lag-points) takes the same 26-28 ms on either: 512.
constexpr int kRefineBlocks = 512;
constexpr int kRefineThreads = 256;
constexpr int kRefineItems = 2048;    // a sweep with few flagged lag-points is cut in about this many work items
constexpr int kRefineMaxChunks = 64;

struct FinalizeArgs {
    const double* partials;
    int n_groups;  // number of partial slabs (tile groups x point groups)
    long long n_slots;
    const long long* out_index;  // C-order raveled lag index of each slot, or -1 (padding)
    long long lag_begin;
    double* out;  // [lag_end - lag_begin]
    int residus;          // 1: np.std((A - B) / sqrt(A)) over ALL grid points (alignment.py:544-547)
    long long n_required;  // residus: number of grid points G; fewer contributions -> NaN (no mask in that method)
    // multi-GPU point sharding: instead of the coefficient, write the six sums of this rank's groups to
    // sums_out[k * sums_stride + sums_off + slot] (all-reduced over the ranks, then finalised by a second call with
    // n_groups = 1 and partials = the reduced sums)
    double* sums_out;
    long long sums_stride, sums_off;
    long long part_stride;  // distance between the six sums of a slab (n_slots, or sums_stride when reading reduced sums)
    RefineArgs refine;
    long long* refine_count;  // device counters (diagnostics), or null: [0] re-evaluated lag-points, [1] lag-points that
                              // were flagged but kept their one-pass value (fix_slab below; there is no cap)
    // the extra slab of a launch with noise-decided samples (k_border_fix / k_parity_fix / k_tap_fix), [kNumSums][n_slots],
    // or null.  With refine.fix_slab set (the fix kernels run a second time, about the slot pivots) every flagged
    // lag-point is re-evaluated.  Without it, a lag-point whose entries are all zero had nothing taken out or put in and
    // is re-evaluated like any other; one with a correction keeps its one-pass value (the re-evaluation walks the grid
    // without the lists of those samples) and is counted in refine_count[1] when it was flagged.
    const double* fix_slab;
};
constexpr int kFinLanes = 16;  // threads per lag slot in k_finalize
constexpr int kFinSlots = 16;  // lag slots per block: 256-thread blocks, 16 of them per 256-lag batch -- a sweep of two
                               // batches (one GPU's share of the headline at N = 8) still spreads over 32 CUs
constexpr int kFinThreads = kFinSlots * kFinLanes;

// flagged slots in slot order (ONE block of kListThreads threads: deterministic), their number, and the number of chunks each
// one's walk over the tile list is cut in: few flagged lag-points -> many chunks each, so that the re-evaluation still
// fills the chip
constexpr int kListThreads = 1024;
__device__ void refine_list_block(const RefineArgs& r, long long n_slots, long long* refine_count) {
    __shared__ int wave_n[kListThreads / 64];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kPre = 4;   // flags fetched ahead per thread: the loads (other XCDs wrote them) overlap instead of queueing
    for (long long c0 = 0; c0 < n_slots; c0 += (long long)kPre * kListThreads) {
        int f[kPre];
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const long long slot = c0 + (long long)q * kListThreads + threadIdx.x;
            f[q] = slot < n_slots ? ((volatile const int*)r.flags)[slot] : 0;
        }
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const long long s0 = c0 + (long long)q * kListThreads;
            if (s0 >= n_slots) break;  // (uniform)
            const unsigned long long m = __ballot(f[q] != 0);
            if (lane == 0) wave_n[wave] = __popcll(m);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wave; ++w) off += wave_n[w];
            if (f[q]) r.list[off + __popcll(m & ((1ull << lane) - 1ull))] = (int)(s0 + threadIdx.x);
            __syncthreads();
            if (threadIdx.x == 0) {
                int t = 0;
                for (int w = 0; w < kListThreads / 64; ++w) t += wave_n[w];
                base += t;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        const int n = base;
        r.head[0] = n;
        int chunks = n > 0 ? kRefineItems / n : 1;
        chunks = chunks < 1 ? 1 : (chunks > kRefineMaxChunks ? kRefineMaxChunks : chunks);
        r.head[1] = chunks;
        // (head[2] is unused; k_refine's ticket is head[3], which k_refine itself leaves at zero -- ADVICE r05)
        if (refine_count && n > 0) atomicAdd((unsigned long long*)refine_count, (unsigned long long)n);
    }
}

__global__ void __launch_bounds__(kFinThreads) k_finalize(const FinalizeArgs a) {
    // kFinLanes threads per slot each add every kFinLanes-th slab, then one adds them in order
    __shared__ double red[kFinLanes - 1][kNumSums][kFinSlots];
    const int ls = threadIdx.x % kFinSlots, j = threadIdx.x / kFinSlots;
    const long long slot = (long long)blockIdx.x * kFinSlots + ls;
    double s[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) s[k] = 0.0;
    if (slot < a.n_slots) {
        for (int g = j; g < a.n_groups; g += kFinLanes) {
            const double* p = a.partials + (size_t)g * kNumSums * a.part_stride + slot;
#pragma unroll
            for (int k = 0; k < kNumSums; ++k) s[k] += p[(size_t)k * a.part_stride];
        }
    }
    if (j > 0) {
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) red[j - 1][k][ls] = s[k];
    }
    __syncthreads();
    if (j == 0 && slot < a.n_slots) {
        for (int g = 0; g < kFinLanes - 1; ++g) {
#pragma unroll
            for (int k = 0; k < kNumSums; ++k) s[k] += red[g][k][ls];
        }
        int flag = 0;
        if (a.sums_out) {
#pragma unroll
            for (int k = 0; k < kNumSums; ++k) a.sums_out[(size_t)k * a.sums_stride + a.sums_off + slot] = s[k];
        } else {
            const long long idx = a.out_index[slot];
            if (idx >= 0) {
                const double n = s[0];
                double r = __builtin_nan("");
                if (a.residus) {
                    if (n == (double)a.n_required) {
                        const double m = s[2] / n;
                        r = sqrt(fmax(s[4] / n - m * m, 0.0));
                    }
                } else if (n > 1.0) {  // (one sample: the reference's centred sums are 0 / sqrt(0 * 0) = NaN, exactly)
                    const double cov = s[5] - s[1] * s[2] / n;
                    const double va = s[3] - s[1] * s[1] / n;
                    const double vb = s[4] - s[2] * s[2] / n;
                    r = cov / sqrt(va * vb);
                    // (negated comparisons: a NaN or non-positive variance is flagged too)
                    flag = !(va > 0.0) || !(vb > 0.0) || !(s[3] <= a.refine.cond * va) || !(s[4] <= a.refine.cond * vb);
                    if (a.refine.enabled && flag && a.fix_slab && !a.refine.fix_slab) {
                        bool corrected = false;
#pragma unroll
                        for (int k = 0; k < kNumSums; ++k) corrected |= a.fix_slab[(size_t)k * a.n_slots + slot] != 0.0;
                        if (corrected) {
                            flag = 0;
                            if (a.refine_count) atomicAdd((unsigned long long*)a.refine_count + 1, 1ull);
                        }
                    }
                    if (a.refine.enabled && flag) {
                        a.refine.slot_pivots[slot] = s[1] / n;
                        a.refine.slot_pivots[a.n_slots + slot] = s[2] / n;
                    }
                }
                a.out[idx - a.lag_begin] = r;
            }
        }
        if (a.refine.enabled) a.refine.flags[slot] = flag;
    }
}
// (Listing the flagged slots by "the block of k_finalize that finishes last" was tried and is SLOWER than this one-block
// kernel: the device-scope fence it needs writes the XCD's L2 back -- k_finalize 10 -> 34 us on the headline.)
__global__ void __launch_bounds__(kListThreads) k_refine_list(const RefineArgs r, long long n_slots, long long* refine_count) {
    refine_list_block(r, n_slots, refine_count);
}

// one work item = (flagged lag-point, chunk of the tile list): six sums about the lag-point's own pivots over the chunk's
// compacted points, with the arithmetic of the sweep's per-point path (point_lag on a zeroed accumulator hands back
// (valid, a - pivot, sample - pivot)); fixed thread -> point assignment and reduction tree
template <int MODE, int ORDER, typename TS>
__device__ void refine_item(const RefineArgs& r, long long n_slots, int slot, int chunk, int n_chunks, double* out6,
                            double (*sh)[kRefineThreads]) {
    constexpr bool ROUND = MODE != MODE_TRANSLATE;
    double px0 = 0.0, py0 = 0.0;
    H9 hm;
#pragma unroll
    for (int k = 0; k < 9; ++k) hm.h[k] = 0.0;
    if (MODE == MODE_TRANSLATE) {
        px0 = r.lane_params[slot];
        py0 = r.lane_params[n_slots + slot];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) hm.h[k] = r.lane_params[(long long)k * n_slots + slot];
    }
    const TS* __restrict__ img = (const TS*)r.img;
    const double wmax = (double)(r.W - 1), hmax = (double)(r.H - 1), pivot_b = r.pivots[1];
    const double pa = r.slot_pivots[slot], pb = r.slot_pivots[n_slots + slot];
    const int n_list = (int)r.tile_info[0];
    double s[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) s[k] = 0.0;
    for (int tl = chunk; tl < n_list; tl += n_chunks) {
        const int tile = r.tile_list[tl];
        const int cnt = r.tile_count[tile];
        const Pt* __restrict__ pts = r.pts + (size_t)tile * kTilePts;
        for (int p = threadIdx.x; p < cnt; p += kRefineThreads) {
            const Pt pt = pts[p];
            Acc t = {0, 0.0, 0.0, 0.0, 0.0, 0.0};
            point_lag<MODE, ORDER, TS, false, ROUND, false>(t, 0u, img, 0, 0, 0, r.W, r.H, wmax, hmax, px0, py0, 0.0, 0.0, hm,
                                                            r.car_inv, pt.b0, pt.b1, pt.a, pt.pad, pivot_b);
            if (t.n) {
                const double da = t.a - pa, db = t.b - pb;
                s[0] += 1.0;
                s[1] += da;
                s[2] += db;
                s[3] = fma(da, da, s[3]);
                s[4] = fma(db, db, s[4]);
                s[5] = fma(da, db, s[5]);
            }
        }
    }
    for (int k = 0; k < kNumSums; ++k) {
        sh[0][threadIdx.x] = s[k];
        __syncthreads();
        for (int o = kRefineThreads / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out6[k] = sh[0][0];
        __syncthreads();
    }
}
template <int MODE, typename TS>
__device__ void refine_order(const RefineArgs& r, long long n_slots, int slot, int chunk, int n_chunks, double* out6,
                             double (*sh)[kRefineThreads]) {
    if (r.order == 2) refine_item<MODE, 2, TS>(r, n_slots, slot, chunk, n_chunks, out6, sh);
    else if (r.order == 1) refine_item<MODE, 1, TS>(r, n_slots, slot, chunk, n_chunks, out6, sh);
    else refine_item<MODE, ORDER_RT, TS>(r, n_slots, slot, chunk, n_chunks, out6, sh);
}
template <typename TS>
__device__ void refine_mode(const RefineArgs& r, long long n_slots, int slot, int chunk, int n_chunks, double* out6,
                            double (*sh)[kRefineThreads]) {
    switch (r.mode) {
        case MODE_TRANSLATE: refine_order<MODE_TRANSLATE, TS>(r, n_slots, slot, chunk, n_chunks, out6, sh); break;
        case MODE_HOMOGRAPHY: refine_order<MODE_HOMOGRAPHY, TS>(r, n_slots, slot, chunk, n_chunks, out6, sh); break;
        case MODE_HOMOGRAPHY_SERIES: refine_order<MODE_HOMOGRAPHY_SERIES, TS>(r, n_slots, slot, chunk, n_chunks, out6, sh); break;
        default: refine_order<MODE_CAR, TS>(r, n_slots, slot, chunk, n_chunks, out6, sh); break;
    }
}
__global__ void __launch_bounds__(kRefineThreads) k_refine(const RefineArgs r, long long n_slots) {
    __shared__ double sh[1][kRefineThreads];
    __shared__ int s_last;
    const int n = r.head[0], n_chunks = r.head[1];  // (uniform; every wave leaves when there is nothing flagged)
    if (n == 0) return;
    const long long items = (long long)n * n_chunks;
    for (long long w = blockIdx.x; w < items; w += gridDim.x) {
        const int slot = r.list[w / n_chunks], chunk = (int)(w % n_chunks);
        double* out6 = r.partial + (size_t)w * kNumSums;
        if (r.small_f32) refine_mode<float>(r, n_slots, slot, chunk, n_chunks, out6, sh);
        else refine_mode<double>(r, n_slots, slot, chunk, n_chunks, out6, sh);
    }
    // the block that finishes last adds the chunks of every flagged lag-point in chunk order and writes the corrected
    // two-pass coefficient
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd((unsigned int*)r.head + 3, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int e = threadIdx.x; e < n; e += kRefineThreads) {
        double s[kNumSums];
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) s[k] = 0.0;
        for (int c = 0; c < n_chunks; ++c) {
            const volatile double* p = r.partial + ((size_t)e * n_chunks + c) * kNumSums;
#pragma unroll
            for (int k = 0; k < kNumSums; ++k) s[k] += p[k];
        }
        const int slot = r.list[e];
        if (r.fix_slab) {
#pragma unroll
            for (int k = 0; k < kNumSums; ++k) s[k] += r.fix_slab[(size_t)k * n_slots + slot];
        }
        const double cnt = s[0];
        double res = __builtin_nan("");
        if (cnt > 1.0) {
            const double cov = s[5] - s[1] * s[2] / cnt;
            const double va = s[3] - s[1] * s[1] / cnt;
            const double vb = s[4] - s[2] * s[2] / cnt;
            res = cov / sqrt(va * vb);
        }
        r.out[r.out_index[slot] - r.lag_begin] = res;
    }
    if (threadIdx.x == 0) r.head[3] = 0;  // the ticket, for the next launch
}

}  // namespace coreg
