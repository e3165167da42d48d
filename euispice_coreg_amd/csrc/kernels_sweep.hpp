// Part of csrc/kernels.hpp (included from there in order; round 6 split by concern, no behaviour change): THE hot kernel: hand-issued LDS gathers (Taps), gather_o2 / gather_o3, point_lag, tile_points (scalar point prefetch, runs), k_sweep.
#pragma once
namespace coreg {
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

}  // namespace coreg
