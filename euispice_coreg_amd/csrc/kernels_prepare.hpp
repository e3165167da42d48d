// Part of csrc/kernels.hpp (included from there in order; round 6 split by concern, no behaviour change): everything before the sweep: pivots, FITS data units decoded on the device, thresholds, the once-only resample, k_precompute (culling, compaction, chunk sums, run flags), k_tile_list.
#pragma once
namespace coreg {
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

}  // namespace coreg
