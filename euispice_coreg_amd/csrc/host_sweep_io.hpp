// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): begin / end of a sweep call: output buffers, timing events and stats, the plan's upload, grid-shared (sums) bookkeeping.
#pragma once
namespace {
int collect_stats(coreg_handle* h);

int begin_sweep(coreg_handle* h, long long n_out, double* corr_out, int out_on_device, double** out_dev) {
    // timings of a still-uncollected device-output sweep are dropped (its events are re-recorded below): starting the
    // next sweep never waits for the previous one
    h->stats_pending = false;
    // sums of an earlier point-sharded sweep must not outlive it: a sweep that returns early (empty slice, no launch)
    // would otherwise leave them for coreg_copy_sums / coreg_finalize_sums to pick up
    h->pending_fin.clear();
    h->sums_slots = 0;
    h->pending_n_out = 0;
    h->tap_last[0] = h->tap_last[1] = h->tap_last[2] = 0;  // (coreg_last_tap_fix speaks of THIS sweep)
    // A sweep that failed between upload_plan and end_sweep leaves its prologue armed and may have enqueued kernels that
    // still read its pinned plan slot: forget the prologue, and let everything it enqueued finish before that slot (it
    // was never handed on) is written again.
    std::memset(&h->pending_prologue, 0, sizeof(h->pending_prologue));
    if (h->plan_open) {
        h->plan_open = false;
        RETCHK(bind_device(h));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (!h->small.p) return fail(h, COREG_ESTATE, "coreg_set_small has not been called");
    if (!h->ref.p) return fail(h, COREG_ESTATE, "no reference image on the target grid");
    if (!corr_out && n_out > 0) return fail(h, COREG_EINVAL, "corr_out is null");
    h->ev_sweep_used = 0;
    h->ev_pre_used = 0;
    std::memset(&h->stats, 0, sizeof(h->stats));
    h->stats.small_is_f32 = h->small_f32 ? 1 : 0;
    h->stats.n_grid_points = (long long)h->gW * h->gH;
    h->stats.n_lags = n_out;
    if (out_on_device) {
        *out_dev = corr_out;
    } else {
        HIPCHK(h->out_dev.reserve((size_t)std::max<long long>(n_out, 1) * sizeof(double)));
        *out_dev = h->out_dev.as<double>();
    }
    // (no start event of its own: the opening event of the first k_precompute launch is the sweep's start, collect_stats)
    // (the output is NaN-initialised by the prologue part of the first k_precompute launch, or by fill_nan on the
    // paths that launch nothing)
    return COREG_OK;
}

int fill_nan(coreg_handle* h, double* out_dev, long long n_out) {
    if (n_out > 0) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, h->stream, out_dev,
                           (long long)n_out, std::numeric_limits<double>::quiet_NaN());
        HIPCHK(hipGetLastError());
    }
    return COREG_OK;
}

int collect_stats(coreg_handle* h) {
    if (!h->stats_pending) return COREG_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stats_pending = false;
    if (h->tile_info.p && h->stats.n_sweep_launches > 0) {
        long long info[3] = {0, 0, 0};
        HIPCHK(hipMemcpy(info, h->tile_info.p, sizeof(info), hipMemcpyDeviceToHost));
        h->stats.n_active_points = info[1];
    }
    float ms = 0.f;
    for (size_t i = 0; i < h->ev_sweep_used; ++i) {
        HIPCHK(hipEventElapsedTime(&ms, h->ev_sweep[i].a, h->ev_sweep[i].b));
        h->stats.sweep_kernel_ms += ms;
    }
    for (size_t i = 0; i < h->ev_pre_used; ++i) {
        const EventPair& e = h->ev_pre[i];
        if (e.b_is_next_sweep && e.next_sweep_index >= h->ev_sweep_used) continue;  // (no sweep launch followed)
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b_is_next_sweep ? h->ev_sweep[e.next_sweep_index].a : e.b));
        h->stats.precompute_ms += ms;
    }
    if (h->ev_pre_used > 0) {
        HIPCHK(hipEventElapsedTime(&ms, h->ev_pre[0].a, h->ev_t1));
        h->stats.total_gpu_ms = ms;
    }
    return COREG_OK;
}

// Host output: copy back and wait.  Device output: return at once -- the sweep is stream-ordered work like any other
// (a following collective on the same stream sees the results); timings are gathered when coreg_last_stats asks.
int end_sweep(coreg_handle* h, long long n_out, double* corr_out, int out_on_device, double* out_dev) {
    h->ev_t1 = h->ev_end[h->plan_slot];  // this sweep's end: statistics, and the guard of its plan staging slot
    HIPCHK(hipEventRecord(h->ev_t1, h->stream));
    h->plan_slot ^= 1;
    h->plan_open = false;
    if (!out_on_device && n_out > 0)
        HIPCHK(hipMemcpyAsync(corr_out, out_dev, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    h->stats_pending = true;  // (timings and the kept-point count are gathered when coreg_last_stats asks: collect_stats)
    // host output: the values must be there on return; device output: the sweep stays stream-ordered work
    trace("end_sweep: everything issued");
    if (!out_on_device) HIPCHK(hipStreamSynchronize(h->stream));
    trace("end_sweep: map on the host");
    return COREG_OK;
}

long long lds_window_elems(const coreg_handle* h) {
    const size_t lds_min = (size_t)(kPointGroups - 1) * kNumSums * kBlock * sizeof(double);
    return (long long)(std::max(lds_min, (size_t)h->opt_lds_bytes) / sizeof(double));
}

// The concatenated per-launch lag parameters / output indices go to page-locked memory; the FIRST k_precompute launch of
// the sweep fetches them from there and NaN-initialises the output (PrologueArgs: no host sync, no DMA-engine copy, no
// launch of its own between the kernels).
int upload_plan(coreg_handle* h, const std::vector<double>& params, const std::vector<long long>& outidx,
                double* out_dev, long long n_out) {
    const size_t bytes = params.size() * sizeof(double) + outidx.size() * sizeof(long long);
    HIPCHK(h->lane_params.reserve(params.size() * sizeof(double)));
    HIPCHK(h->out_index.reserve(outidx.size() * sizeof(long long)));
    PinBuf& pin = h->pin_plan[h->plan_slot];
    HIPCHK(hipEventSynchronize(h->ev_end[h->plan_slot]));  // the sweep before last (same slot) has ended
    h->plan_open = true;  // (closed by end_sweep; begin_sweep cleans up after a sweep that never got there)
    HIPCHK(pin.reserve(bytes));
    std::memcpy(pin.p, params.data(), params.size() * sizeof(double));
    std::memcpy((char*)pin.p + params.size() * sizeof(double), outidx.data(), outidx.size() * sizeof(long long));
    void* src_dev = nullptr;
    HIPCHK(hipHostGetDevicePointer(&src_dev, pin.p, 0));
    PrologueArgs& p = h->pending_prologue;
    p.src = (const double*)src_dev;
    p.dst_params = h->lane_params.as<double>();
    p.n_params = (long long)params.size();
    p.dst_outidx = h->out_index.as<long long>();
    p.n_outidx = (long long)outidx.size();
    p.out = out_dev;
    p.n_out = n_out;
    HIPCHK(h->counters.reserve(8 * sizeof(long long)));
    p.refine_count = h->counters.as<long long>();
    return COREG_OK;
}

// point-sharded sweep (coreg_set_option "shard_world" > 1): room for the six sums of every slot of every launch, and a
// private copy of the slots' output indices for coreg_finalize_sums
int prepare_sharded(coreg_handle* h, size_t total_slots, long long n_out, long long lag_begin) {
    h->pending_fin.clear();
    h->sums_slots = 0;
    h->pending_n_out = n_out;
    (void)lag_begin;
    if (h->opt_shard_world <= 1) return COREG_OK;
    h->sums_slots = (long long)total_slots;
    HIPCHK(h->sums.reserve(std::max<size_t>(1, total_slots) * kNumSums * sizeof(double)));
    HIPCHK(h->fin_outidx.reserve(std::max<size_t>(1, total_slots) * sizeof(long long)));
    // (from the pinned plan staging: the device copy of the output indices is only written by the prologue part of the
    // sweep's first k_precompute launch, which has not been enqueued yet)
    const PrologueArgs& pr = h->pending_prologue;
    if (!pr.src || (size_t)pr.n_outidx < total_slots) return fail(h, COREG_ESTATE, "prepare_sharded: no plan staged");
    HIPCHK(hipMemcpyAsync(h->fin_outidx.p, (const char*)h->pin_plan[h->plan_slot].p + (size_t)pr.n_params * sizeof(double),
                          total_slots * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    return COREG_OK;
}

}  // namespace
