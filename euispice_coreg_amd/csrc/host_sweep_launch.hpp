// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): launch_sweep: k_sweep dispatch by (mode, order, pixel type, pitch), the fix slab, k_finalize, re-evaluation of flagged lag-points.
#pragma once
namespace {
int launch_sweep(coreg_handle* h, int mode, int order, int method, const double* params_dev,
                 const long long* outidx_dev, int n_batches, int n_tiles, long long lag_begin, double* out_dev,
                 const LaunchU* car_inv = nullptr, const BorderFix* fix = nullptr, long long sums_off = 0,
                 int pitch_sel = 0) {
    const long long n_slots = (long long)n_batches * kBlock;
    const int n_groups = pick_groups(h, n_batches, n_tiles);
    const bool sharded = h->opt_shard_world > 1;
    const int g_per = sharded ? n_groups / (int)h->opt_shard_world : n_groups;  // groups swept by this launch
    const int g_lo = sharded ? g_per * (int)h->opt_shard_rank : 0;
    // (the border correction is a property of the lag-point, not of a share of the grid: rank 0 carries it)
    const bool fixing = fix && (!fix->items.empty() || fix->tap_segs > 0) && (!sharded || h->opt_shard_rank == 0);
    HIPCHK(h->partials.reserve((size_t)(g_per + (fixing ? 1 : 0)) * kNumSums * n_slots * sizeof(double)));

    SweepArgs a;
    a.img = h->small.p;
    a.W = h->sW;
    a.H = h->sH;
    a.pts = h->pts.as<Pt>();
    a.tile_count = h->tile_count.as<int>();
    a.tile_list = h->tile_list.as<int>();
    a.tile_cum = h->tile_cum.as<int>();
    a.group_first = h->group_first.as<int>();
    a.tile_info = h->tile_info.as<long long>();
    a.tile_bbox = h->tile_bbox.as<double>();
    a.lane_params = params_dev;
    a.n_slots = n_slots;
    a.n_batches = n_batches;
    a.n_groups = n_groups;
    a.group_lo = g_lo;
    a.partials = h->partials.as<double>();
    a.pivots = h->pivots.as<double>();
    a.use_lds = h->opt_use_lds ? 1 : 0;
    a.clean_path = h->opt_clean_path ? 1 : 0;
    // the dynamic LDS also carries the end-of-kernel point-group reduction: (kPointGroups-1) x 6 x 256 doubles
    const size_t lds_min = (size_t)(kPointGroups - 1) * kNumSums * kBlock * sizeof(double);
    const size_t lds_bytes = std::max(lds_min, a.use_lds ? (size_t)h->opt_lds_bytes : 0);
    a.lds_elems = (int)(lds_bytes / sizeof(double));
    std::memset(&a.car_inv, 0, sizeof(a.car_inv));
    if (car_inv) a.car_inv = *car_inv;
    a.car_inv.order_rt = order;
    a.car_inv.h_incr = (int)h->opt_h_incr;

    const dim3 grid((unsigned)((long long)g_per * n_batches)), block(kSweepThreads);
    RETCHK(join_small(h));  // the first kernel of the call that reads the image to align
    trace("launch_sweep: launching k_sweep");
    EventPair* ev = next_event(h, h->ev_sweep, h->ev_sweep_used);
    if (!ev) return fail(h, COREG_EHIP, "hipEventCreate failed");
#define SWP(M, O, TS, R, Q, P)                                                                                       \
    do {                                                                                                              \
        {                                                                                                             \
            /* per instantiation and device: raise the dynamic-LDS limit once, not per launch (handles of several  */ \
            /* threads share the function attribute, hence the lock)                                                */ \
            static size_t attr_bytes[kMaxDevices] = {0};                                                              \
            std::lock_guard<std::mutex> lock(g_attr_mutex);                                                           \
            size_t& ab = attr_bytes[h->device % kMaxDevices];                                                         \
            if (lds_bytes > 48 * 1024 && lds_bytes > ab) {                                                            \
                HIPCHK(hipFuncSetAttribute((const void*)(k_sweep<M, O, TS, R, Q, P>),                                 \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));              \
                ab = lds_bytes;                                                                                       \
            }                                                                                                         \
        }                                                                                                             \
        HIPCHK(hipEventRecord(ev->a, h->stream));                                                                     \
        hipLaunchKernelGGL((k_sweep<M, O, TS, R, Q, P>), grid, block, lds_bytes, h->stream, a);                       \
        HIPCHK(hipEventRecord(ev->b, h->stream));                                                                     \
    } while (0)
#define SW(M, O, TS, R, Q) SWP(M, O, TS, R, Q, 0)
#define SW_Q(M, O, TS, R)                                        \
    do {                                                         \
        if (method == COREG_METHOD_RESIDUS) SW(M, O, TS, R, true); \
        else SW(M, O, TS, R, false);                             \
    } while (0)
#define SW_T(M, O, R)                           \
    do {                                        \
        if (h->small_f32) SW_Q(M, O, float, R); \
        else SW_Q(M, O, double, R);             \
    } while (0)
    // TRANSLATE = Carrington (float64 samples); HOMOGRAPHY[_SERIES] = helioprojective (samples rounded to float32)
    if (mode == MODE_TRANSLATE && order == 2 && h->small_f32 && method != COREG_METHOD_RESIDUS && pitch_sel > 0) {
        // the common Carrington sweep with a compile-time window pitch (pick_pitch)
        switch (pitch_sel) {
            case 89: SWP(MODE_TRANSLATE, 2, float, false, false, 89); break;
            case 121: SWP(MODE_TRANSLATE, 2, float, false, false, 121); break;
            case 153: SWP(MODE_TRANSLATE, 2, float, false, false, 153); break;
            case 185: SWP(MODE_TRANSLATE, 2, float, false, false, 185); break;
            case 217: SWP(MODE_TRANSLATE, 2, float, false, false, 217); break;
            default: SW(MODE_TRANSLATE, 2, float, false, false); break;
        }
    } else if (mode == MODE_TRANSLATE && order == 2 && !h->small_f32 && method != COREG_METHOD_RESIDUS &&
               (pitch_sel == 89 || pitch_sel == 121 || pitch_sel == 153)) {
        // the same with float64 pixels (values that are not float32-exact): the three smallest pitches
        if (pitch_sel == 89) SWP(MODE_TRANSLATE, 2, double, false, false, 89);
        else if (pitch_sel == 121) SWP(MODE_TRANSLATE, 2, double, false, false, 121);
        else SWP(MODE_TRANSLATE, 2, double, false, false, 153);
    } else if ((mode == MODE_HOMOGRAPHY_SERIES || mode == MODE_HOMOGRAPHY) && order == 2 && h->small_f32 &&
               method != COREG_METHOD_RESIDUS && (pitch_sel == 89 || pitch_sel == 121)) {
        // the common helioprojective sweeps likewise
        if (mode == MODE_HOMOGRAPHY_SERIES) {
            if (pitch_sel == 89) SWP(MODE_HOMOGRAPHY_SERIES, 2, float, true, false, 89);
            else SWP(MODE_HOMOGRAPHY_SERIES, 2, float, true, false, 121);
        } else {
            if (pitch_sel == 89) SWP(MODE_HOMOGRAPHY, 2, float, true, false, 89);
            else SWP(MODE_HOMOGRAPHY, 2, float, true, false, 121);
        }
    } else if (mode == MODE_TRANSLATE && order == 3 && h->small_f32 && method != COREG_METHOD_RESIDUS &&
               (pitch_sel == 89 || pitch_sel == 121 || pitch_sel == 153)) {
        // the cubic Carrington sweep with a compile-time window pitch
        if (pitch_sel == 89) SWP(MODE_TRANSLATE, 3, float, false, false, 89);
        else if (pitch_sel == 121) SWP(MODE_TRANSLATE, 3, float, false, false, 121);
        else SWP(MODE_TRANSLATE, 3, float, false, false, 153);
    } else if (mode == MODE_TRANSLATE) {
        if (order == 2) SW_T(MODE_TRANSLATE, 2, false);
        else if (order == 1) SW_T(MODE_TRANSLATE, 1, false);
        else if (order == 3) SW_T(MODE_TRANSLATE, 3, false);
        else SW_T(MODE_TRANSLATE, ORDER_RT, false);
    } else if (mode == MODE_CAR) {
        if (order == 2) SW_T(MODE_CAR, 2, true);
        else if (order == 1) SW_T(MODE_CAR, 1, true);
        else SW_T(MODE_CAR, ORDER_RT, true);
    } else if (mode == MODE_HOMOGRAPHY_SERIES && (order == 1 || order == 2 || order == 3)) {
        if (order == 2) SW_T(MODE_HOMOGRAPHY_SERIES, 2, true);
        else if (order == 3) SW_T(MODE_HOMOGRAPHY_SERIES, 3, true);
        else SW_T(MODE_HOMOGRAPHY_SERIES, 1, true);
    } else {
        if (order == 2) SW_T(MODE_HOMOGRAPHY, 2, true);
        else if (order == 1) SW_T(MODE_HOMOGRAPHY, 1, true);
        else if (order == 3) SW_T(MODE_HOMOGRAPHY, 3, true);
        else SW_T(MODE_HOMOGRAPHY, ORDER_RT, true);
    }
#undef SWP
#undef SW_Q
#undef SW_T
#undef SW
    HIPCHK(hipGetLastError());
    h->stats.n_sweep_launches++;
    h->stats.used_lds = a.use_lds;

    // the noise-decided samples of this launch as kernel arguments (FixLaunch): built on every rank of a
    // grid-shared sweep (the re-evaluation of a flagged lag-point runs on every rank), run here -- about the global
    // pivots, into the extra slab -- on the rank that carries the correction
    FixLaunch fl;
    fl.small_f32 = h->small_f32;
    if (fix && (!fix->items.empty() || fix->tap_segs > 0)) {
        BorderFixArgs b = {};
        b.img = h->small.p;
        b.W = h->sW;
        b.H = h->sH;
        b.ref = h->ref.p;
        b.ref_f32 = h->ref_dtype == COREG_F32 ? 1 : 0;
        b.gw = h->gW;
        b.order = order;
        b.round_f32 = mode == MODE_TRANSLATE ? 0 : 1;
        b.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
        b.pivots = h->pivots.as<double>();
        b.n_slots = n_slots;
        for (const BorderFix::Item& it : fix->items) {
            b.slot = it.slot;
            b.dropped = h->border_dev.as<int>() + it.first;
            b.n_dropped = it.n;
            b.hom = params_dev;  // SoA [9][n_slots]: the (snapped) map of the slot gives the sample coordinates
            if (it.flags_off >= 0) {
                // odd spline order: re-decide the tap set of every pixel of this lag-point (k_parity_fix), after
                // k_border_fix has set the slab entry (same stream)
                ParityFixArgs p = {};
                p.img = b.img;
                p.W = b.W;
                p.H = b.H;
                p.ref = b.ref;
                p.ref_f32 = b.ref_f32;
                p.flags = h->border_flags.as<unsigned char>() + it.flags_off;
                p.gw = h->gW;
                p.gh = h->gH;
                p.order = order;
                p.round_f32 = b.round_f32;
                p.residus = b.residus;
                p.pivots = b.pivots;
                p.hom = params_dev;
                p.n_slots = n_slots;
                p.slot = it.slot;
                p.n_partial = 256;
                if (h->fix_partial.reserve((size_t)p.n_partial * kNumSums * sizeof(double)) != hipSuccess)
                    return fail(h, COREG_EHIP, "hipMalloc failed (parity fix)");
                p.partial = h->fix_partial.as<double>();
                fl.parity.push_back(p);
            }
            if (it.n == 0) continue;
            fl.border.push_back(b);
        }
        if (fix->tap_segs > 0) {
            TapFixArgs t = fix->tap;
            t.img = b.img;
            t.W = b.W;
            t.H = b.H;
            t.ref = b.ref;
            t.ref_f32 = b.ref_f32;
            t.gw = h->gW;
            t.order = order;
            t.round_f32 = b.round_f32;
            t.residus = b.residus;
            t.pivots = b.pivots;
            t.hom = params_dev;
            t.n_slots = n_slots;
            fl.tap = t;
            fl.have_tap = true;
            fl.tap_segs = fix->tap_segs;
            fl.tap_count = fix->tap_count;
            fl.tap_mode = fix->tap_mode;
        }
    }
    if (fixing) {
        // one more slab: zero, except minus the dropped border pixels' totals at the identity lag's slot
        double* slab = h->partials.as<double>() + (size_t)g_per * kNumSums * n_slots;
        HIPCHK(hipMemsetAsync(slab, 0, (size_t)kNumSums * n_slots * sizeof(double), h->stream));
        RETCHK(launch_fix_kernels(h, fl, slab, nullptr, nullptr));
    }

    FinalizeArgs f = {};
    // ill-conditioned lag-points are flagged by k_finalize and re-evaluated about their own means (kernels.hpp:
    // RefineArgs) -- not for method 'residus' (another statistic) and not the lag-points whose noise-decided samples
    // were taken out of (put into) the sums by the extra slab: those keep their one-pass value (FinalizeArgs.fix_slab).
    // Grid shares across GPUs: the flags can only come from the REDUCED sums, so the re-evaluation is run by
    // coreg_finalize_sums, on every rank, over the whole grid (with the second run of this launch's fix kernels).
    const bool refinable = h->opt_refine && method != COREG_METHOD_RESIDUS;
    RETCHK(fill_refine(h, &f.refine, mode, order, params_dev, a.car_inv, n_slots));
    f.refine.enabled = (refinable && !sharded) ? 1 : 0;
    f.fix_slab = fixing ? h->partials.as<double>() + (size_t)g_per * kNumSums * n_slots : nullptr;
    if (fixing && f.refine.enabled) {
        // the re-evaluation of a flagged lag-point of THIS launch needs its noise-decided samples about its own pivots:
        // a second slab, filled between the listing of the flags and k_refine (kernels that leave at once unless the
        // slot is flagged)
        HIPCHK(h->rf_fix_slab.reserve((size_t)kNumSums * n_slots * sizeof(double)));
        HIPCHK(hipMemsetAsync(h->rf_fix_slab.p, 0, (size_t)kNumSums * n_slots * sizeof(double), h->stream));
        f.refine.fix_slab = h->rf_fix_slab.as<double>();
    }
    f.refine_count = h->counters.as<long long>();  // (null before the first plan: no sweep without one)
    f.partials = h->partials.as<double>();
    f.n_groups = g_per + (fixing ? 1 : 0);
    f.part_stride = n_slots;
    f.sums_out = nullptr;
    f.sums_stride = f.sums_off = 0;
    if (sharded) {
        // leave this launch's six sums per slot in h->sums (reserved by the caller for all launches of the sweep)
        f.sums_out = h->sums.as<double>();
        f.sums_stride = h->sums_slots;
        f.sums_off = sums_off;
        coreg_handle::PendingFinalize pf;
        pf.slot_off = sums_off;
        pf.n_slots = n_slots;
        pf.lag_begin = lag_begin;
        pf.outidx_dev = nullptr;  // set by coreg_finalize_sums from fin_outidx
        pf.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
        pf.refine = f.refine;
        pf.refine.enabled = refinable ? 1 : 0;
        pf.replay_precompute = h->last_precompute;
        pf.fixes = fl;  // (the slab is inside the reduced sums; the second run of the fix kernels happens on every rank)
        // a plate-carree sweep has one launch per combination and every launch lists its single samples anew in the
        // handle's buffers: this launch's lists are COPIED (round 6, closes DESIGN 9 open 3 of round 5) so that
        // coreg_finalize_sums can run the fix kernels a second time about the flagged slots' pivots, as FixLaunch lets it
        // do for the one-launch helioprojective sweeps.  Rare path (unrotated maps, single-axis lags): blocking copies.
        if (mode == MODE_CAR && fl.have_tap) {
            auto kept = std::make_shared<KeptTapLists>();
            const size_t nseg = (size_t)fl.tap_segs, cnt = (size_t)fl.tap_count;
            HIPCHK(kept->seg_slot.reserve(std::max<size_t>(nseg, 1) * sizeof(int)));
            HIPCHK(kept->seg_begin.reserve((nseg + 1) * sizeof(int)));
            HIPCHK(kept->pixel.reserve(std::max<size_t>(cnt, 1) * sizeof(unsigned)));
            HIPCHK(kept->xw.reserve(std::max<size_t>(cnt, 1) * sizeof(double)));
            HIPCHK(kept->yw.reserve(std::max<size_t>(cnt, 1) * sizeof(double)));
            HIPCHK(hipMemcpy(kept->seg_slot.p, fl.tap.seg_slot, nseg * sizeof(int), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(kept->seg_begin.p, fl.tap.seg_begin, (nseg + 1) * sizeof(int), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(kept->pixel.p, fl.tap.pixel, cnt * sizeof(unsigned), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(kept->xw.p, fl.tap.xw, cnt * sizeof(double), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(kept->yw.p, fl.tap.yw, cnt * sizeof(double), hipMemcpyDeviceToDevice));
            pf.fixes.tap.seg_slot = kept->seg_slot.as<int>();
            pf.fixes.tap.seg_begin = kept->seg_begin.as<int>();
            pf.fixes.tap.pixel = kept->pixel.as<unsigned>();
            pf.fixes.tap.xw = kept->xw.as<double>();
            pf.fixes.tap.yw = kept->yw.as<double>();
            pf.fixes.kept = kept;
        }
        h->pending_fin.push_back(pf);
    }
    f.n_slots = n_slots;
    f.out_index = outidx_dev;
    f.lag_begin = lag_begin;
    f.out = out_dev;
    f.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
    f.n_required = (long long)h->gW * h->gH;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((n_slots + kFinSlots - 1) / kFinSlots)), dim3(kFinSlots * kFinLanes), 0,
                       h->stream, f);
    HIPCHK(hipGetLastError());
    if (f.refine.enabled && f.refine.fix_slab) {
        hipLaunchKernelGGL(k_refine_list, dim3(1), dim3(kListThreads), 0, h->stream, f.refine, n_slots, h->counters.as<long long>());
        RETCHK(launch_fix_kernels(h, fl, h->rf_fix_slab.as<double>(), f.refine.slot_pivots, f.refine.flags));
        RETCHK(launch_refine(h, f.refine, n_slots, outidx_dev, lag_begin, out_dev, false));
    } else if (f.refine.enabled) {
        RETCHK(launch_refine(h, f.refine, n_slots, outidx_dev, lag_begin, out_dev));
    }
    return COREG_OK;
}

}  // namespace
