// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): noise-decided samples, launch side: work space of the re-evaluation, the fix kernels of one launch.
#pragma once
namespace {
struct BorderFix {  // lag-points of a launch whose border pixels are decided by wcslib's rounding noise
    struct Item {
        long long slot;  // slot of the launch
        int first, n;    // its pixels in h->border_dev: [first, first + n)
        long long flags_off;  // odd spline order: offset of its per-pixel tap-shift flags in h->border_flags, or -1
    };
    std::vector<Item> items;
    std::vector<int> pixels;  // concatenated linear grid indices (host copy of h->border_dev)
    // single samples near an integer coordinate (odd spline orders): device arrays ready for k_tap_fix
    int tap_segs = 0;
    int tap_mode = 0;
    long long tap_count = 0;
    TapFixArgs tap = {};
};

// work space + arguments of the re-evaluation of ill-conditioned lag-points (kernels.hpp: RefineArgs) for a launch of
// n_slots lag slots whose parameters are at params_dev
int fill_refine(coreg_handle* h, RefineArgs* r, int mode, int order, const double* params_dev, const LaunchU& car_inv,
                long long n_slots) {
    HIPCHK(h->rf_flags.reserve((size_t)n_slots * sizeof(int)));
    HIPCHK(h->rf_pivots.reserve((size_t)n_slots * 2 * sizeof(double)));
    HIPCHK(h->rf_list.reserve((size_t)n_slots * sizeof(int)));
    if (!h->rf_head.p) {
        HIPCHK(h->rf_head.reserve(4 * sizeof(int)));
        HIPCHK(hipMemsetAsync(h->rf_head.p, 0, 4 * sizeof(int), h->stream));  // (the two tickets start at zero)
    }
    // work items: (flagged slots) x (chunks per slot) <= max(kRefineItems, n_slots), see refine_list_block
    HIPCHK(h->rf_partial.reserve((size_t)std::max<long long>(kRefineItems, n_slots) * kNumSums * sizeof(double)));
    std::memset(r, 0, sizeof(*r));
    r->cond = std::pow(10.0, (double)h->opt_refine_cond_log10);
    r->mode = mode;
    r->order = order;
    r->small_f32 = h->small_f32 ? 1 : 0;
    r->img = h->small.p;
    r->W = h->sW;
    r->H = h->sH;
    r->pts = h->pts.as<Pt>();
    r->tile_list = h->tile_list.as<int>();
    r->tile_count = h->tile_count.as<int>();
    r->tile_info = h->tile_info.as<long long>();
    r->lane_params = params_dev;
    r->pivots = h->pivots.as<double>();
    r->car_inv = car_inv;
    r->flags = h->rf_flags.as<int>();
    r->slot_pivots = h->rf_pivots.as<double>();
    r->list = h->rf_list.as<int>();
    r->head = h->rf_head.as<int>();
    r->partial = h->rf_partial.as<double>();
    return COREG_OK;
}

// after a k_finalize that has written the flags: list the flagged slots (one block), re-evaluate them and overwrite their
// coefficients (the last block of k_refine).  Two launches, no host round trip; with nothing flagged (the normal case)
// every block leaves at once.
int launch_refine(coreg_handle* h, const RefineArgs& r0, long long n_slots, const long long* outidx_dev,
                  long long lag_begin, double* out_dev, bool list = true) {
    RefineArgs r = r0;
    r.out_index = outidx_dev;
    r.lag_begin = lag_begin;
    r.out = out_dev;
    if (list) hipLaunchKernelGGL(k_refine_list, dim3(1), dim3(kListThreads), 0, h->stream, r, n_slots, h->counters.as<long long>());
    hipLaunchKernelGGL(k_refine, dim3(kRefineBlocks), dim3(kRefineThreads), 0, h->stream, r, n_slots);
    HIPCHK(hipGetLastError());
    return COREG_OK;
}

// the fix kernels of one launch (FixLaunch) into `slab`; slot_pivots / only_flagged: the second run (kernels.hpp:
// BorderFixArgs)
int launch_fix_kernels(coreg_handle* h, const FixLaunch& fl, double* slab, const double* slot_pivots, const int* only_flagged) {
    for (BorderFixArgs b : fl.border) {
        b.slab = slab;
        b.slot_pivots = slot_pivots;
        b.only_flagged = only_flagged;
        if (fl.small_f32) hipLaunchKernelGGL((k_border_fix<float>), dim3(1), dim3(256), 0, h->stream, b);
        else hipLaunchKernelGGL((k_border_fix<double>), dim3(1), dim3(256), 0, h->stream, b);
    }
    for (ParityFixArgs p : fl.parity) {
        p.slab = slab;
        p.slot_pivots = slot_pivots;
        p.only_flagged = only_flagged;
        if (fl.small_f32) hipLaunchKernelGGL((k_parity_fix<float>), dim3(p.n_partial), dim3(256), 0, h->stream, p);
        else hipLaunchKernelGGL((k_parity_fix<double>), dim3(p.n_partial), dim3(256), 0, h->stream, p);
        hipLaunchKernelGGL(k_parity_fix_final, dim3(1), dim3(64), 0, h->stream, p);
    }
    if (fl.have_tap) {
        TapFixArgs t = fl.tap;
        t.slab = slab;
        t.slot_pivots = slot_pivots;
        t.only_flagged = only_flagged;
        const dim3 tg((unsigned)fl.tap_segs), tb(256);
        if (fl.tap_mode == MODE_CAR) {
            if (fl.small_f32) hipLaunchKernelGGL((k_tap_fix<float, MODE_CAR>), tg, tb, 0, h->stream, t);
            else hipLaunchKernelGGL((k_tap_fix<double, MODE_CAR>), tg, tb, 0, h->stream, t);
        } else if (fl.tap_mode == MODE_HOMOGRAPHY_SERIES) {
            if (fl.small_f32) hipLaunchKernelGGL((k_tap_fix<float, MODE_HOMOGRAPHY_SERIES>), tg, tb, 0, h->stream, t);
            else hipLaunchKernelGGL((k_tap_fix<double, MODE_HOMOGRAPHY_SERIES>), tg, tb, 0, h->stream, t);
        } else {
            if (fl.small_f32) hipLaunchKernelGGL((k_tap_fix<float, MODE_HOMOGRAPHY>), tg, tb, 0, h->stream, t);
            else hipLaunchKernelGGL((k_tap_fix<double, MODE_HOMOGRAPHY>), tg, tb, 0, h->stream, t);
        }
    }
    HIPCHK(hipGetLastError());
    return COREG_OK;
}

}  // namespace
