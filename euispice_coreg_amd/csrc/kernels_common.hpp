// Part of csrc/kernels.hpp (included from there in order; round 6 split by concern, no behaviour change): launch constants, argument structs, the per-lag maps (translation, homography, sphere rotation), scipy's spline weights.
#pragma once
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

}  // namespace coreg
