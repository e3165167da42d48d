// Part of csrc/kernels.hpp (included from there in order; round 6 split by concern, no behaviour change): samples decided by wcslib's rounding noise: k_border_fix, k_parity_fix, k_tap_scan[_segments], k_tap_fix.
#pragma once
namespace coreg {
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
                if (a.bounds_only && lr >= 1.0) {
                    // even orders (round 6): the same crossing argument as in the segment loop below, over the whole row --
                    // pixel a.i_lo + j can be within tol of a bound b only if |c0 + s j - b| <= row bound, i.e. j within
                    // w of the chord's crossing t; no integer there for any of the four bounds, and no axis along which
                    // the row is (nearly) invariant: nothing on this row can be on the bounds rule
                    bool none = true;
#pragma unroll
                    for (int b = 0; b < 4 && none; ++b) {
                        const double c0 = b < 2 ? x0 : y0, c1 = b < 2 ? xe : ye, rb = b < 2 ? row_bx : row_by;
                        const double bound = (b & 1) ? (b < 2 ? a.wmax : a.hmax) : 0.0;
                        if (fmin(c0, c1) - rb > bound || fmax(c0, c1) + rb < bound) continue;  // never near this bound
                        const double sl = (c1 - c0) / lr;
                        if (!(fabs(sl) > 1e-3)) {
                            none = false;
                            break;
                        }
                        const double t = (bound - c0) / sl, w = rb / fabs(sl) + 1e-6;
                        none = ceil(t - w) > floor(t + w);
                    }
                    if (none) continue;
                }
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
                // about the crossing.  The integers of that interval -- usually none -- are tested at once; an axis the
                // lag leaves invariant (|s| ~ 0: the whole segment may sit on the bound) still goes to the queue.
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
                    // pixel i0 + j can be within tol of the bound only if |c0 + sl j - bound| <= bulge, i.e. |j - t| <= w: the
                    // INTEGERS of [t - w, t + w] (w ~ 1e-6 px for these maps: almost always none -- the crossing falls
                    // between two pixels -- so that nothing at all is read or mapped for this row)
                    const double t = (bound - c0) / sl, w = bulge / fabs(sl) + 1e-6;
                    const int lo = max(i0, i0 + (int)ceil(t - w)), hi = min(iend_c - 1, i0 + (int)floor(t + w));
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

}  // namespace coreg
