// Part of csrc/kernels.hpp (included from there in order; round 6 split by concern, no behaviour change): k_finalize (fixed-order slab sum, Pearson, conditioning flags) and the re-evaluation of flagged lag-points: k_refine_list, k_refine.
#pragma once
namespace coreg {
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
