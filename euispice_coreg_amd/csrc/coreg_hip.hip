// libcoreg_hip.so -- C ABI (include/coreg_hip.h) of the MI355X alignment sweep.
// Host side: lag table / header shifting / batching (the part of Alignment._find_best_header_parameters,
// hdrshift/alignment.py:613-797, that is not per-pixel work) and the kernel launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <functional>
#include <vector>

#include "fit.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include <atomic>
#include "ricecomp.hpp"
#include "riceenc.hpp"

using namespace coreg;

#define COREG_VERSION "0.1.0"

#include "host_state.hpp"        // buffers, FixLaunch, coreg_handle
#include "host_upload.hpp"       // uploads, FITS decode, reference crop
#include "host_plan.hpp"         // lag batching, tiles, groups, precompute
#include "host_fix_launch.hpp"   // re-evaluation work space, fix kernels of a launch
#include "host_sweep_launch.hpp" // launch_sweep
#include "host_fix_lists.hpp"    // wcslib-decided border pixels / single samples
#include "host_sweep_io.hpp"     // begin / end of a sweep call, stats, plan upload


// =====================================================================================================================
extern "C" {

const char* coreg_version(void) { return "coreg_hip " COREG_VERSION " (gfx950)"; }

int coreg_create(coreg_handle** out, int device) {
    if (!out) return COREG_EINVAL;
    *out = nullptr;
    coreg_handle* h = new (std::nothrow) coreg_handle();
    if (!h) return COREG_ENOMEM;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        delete h;
        return COREG_EHIP;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) {
        delete h;
        return COREG_EINVAL;
    }
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev_end[0]) != hipSuccess ||
        hipEventCreate(&h->ev_end[1]) != hipSuccess || hipEventRecord(h->ev_end[0], h->stream) != hipSuccess ||
        hipEventRecord(h->ev_end[1], h->stream) != hipSuccess ||
        h->pivots.reserve(2 * sizeof(double)) != hipSuccess ||
        hipMemsetAsync(h->pivots.p, 0, 2 * sizeof(double), h->stream) != hipSuccess) {
        h->own_stream = h->stream != nullptr;
        coreg_destroy(h);  // releases whatever was created
        return COREG_EHIP;
    }
    h->own_stream = true;
    std::memset(&h->stats, 0, sizeof(h->stats));
    *out = h;
    return COREG_OK;
}

void coreg_destroy(coreg_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->up_thread.joinable()) {
        {
            std::unique_lock<std::mutex> lk(h->up_m);
            h->up_cv.wait(lk, [&] { return !h->up_busy && !h->up_has; });
            h->up_stop = true;
        }
        h->up_cv.notify_all();
        h->up_thread.join();
    }
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->up_stream) (void)hipStreamSynchronize(h->up_stream);
    for (int k = 0; k < 2; ++k) {
        h->pin_small[k].release();
        if (h->ev_pin_small[k]) (void)hipEventDestroy(h->ev_pin_small[k]);
    }
    DevBuf* bufs[] = {&h->rf_flags, &h->rf_pivots, &h->rf_list, &h->rf_head, &h->rf_partial, &h->small, &h->ref, &h->pivots, &h->red_sum, &h->red_cnt, &h->t_sin_lon, &h->t_cos_lon,
                      &h->t_cos_lat, &h->t_sin_lat, &h->pts, &h->tile_count, &h->tile_list, &h->tile_cum, &h->group_first,
                      &h->tile_info, &h->tile_bbox, &h->counters, &h->lane_params, &h->out_index, &h->partials, &h->out_dev,
                      &h->tmp_img, &h->up_f64, &h->up_flag, &h->up_raw, &h->rice_blob, &h->rice_rand, &h->dec_img, &h->border_dev, &h->sums, &h->fin_outidx, &h->border_flags, &h->fix_partial, &h->rf_fix_slab};
    for (DevBuf* b : bufs) b->release();
    for (int k = 0; k < 2; ++k) {
        h->pin_img[k].release();
        if (h->ev_img[k]) (void)hipEventDestroy(h->ev_img[k]);
    }
    for (auto& e : h->ev_sweep) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto& e : h->ev_pre) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (int k = 0; k < 2; ++k) {
        if (h->ev_end[k]) (void)hipEventDestroy(h->ev_end[k]);
        h->pin_plan[k].release();
    }
    h->pin_info.release();
    h->pin_border.release();
    h->bbox_buf.release();
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    if (h->up_stream) (void)hipStreamDestroy(h->up_stream);
    if (h->ev_small) (void)hipEventDestroy(h->ev_small);
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    h->red_sum_up.release();
    h->red_cnt_up.release();
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* coreg_last_error(const coreg_handle* h) { return h ? h->err.c_str() : "null handle"; }

int coreg_set_stream(coreg_handle* h, void* hip_stream) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    RETCHK(collect_stats(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) HIPCHK(hipStreamDestroy(h->stream));
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return COREG_OK;
}

int coreg_synchronize(coreg_handle* h) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_set_option(coreg_handle* h, const char* name, int64_t value) {
    if (!h || !name) return COREG_EINVAL;
    const std::string n(name);
    if (n == "use_lds") {
        h->opt_use_lds = value ? 1 : 0;
    } else if (n == "clean_path") {
        h->opt_clean_path = value ? 1 : 0;
    } else if (n == "tap_nan_filter") {
        if (value < 0 || value > 2) return fail(h, COREG_EINVAL, "tap_nan_filter must be 0, 1 or 2");
        h->opt_tap_nan_filter = value;
    } else if (n == "overlap_upload") {
        h->opt_overlap_upload = value ? 1 : 0;
    } else if (n == "async_upload") {
        h->opt_async_upload = value ? 1 : 0;
    } else if (n == "refine") {
        h->opt_refine = value ? 1 : 0;
    } else if (n == "refine_cond_log10") {
        if (value < -3 || value > 15) return fail(h, COREG_EINVAL, "refine_cond_log10 must be in [-3, 15]");
        h->opt_refine_cond_log10 = value;
    } else if (n == "refine_max") {
        // round 4's cap on the re-evaluations per block of lag slots; there is no cap any more (accepted, ignored)
        if (value < 0 || value > 16) return fail(h, COREG_EINVAL, "refine_max must be in [0, 16]");
    } else if (n == "tile_w") {
        if (value != 0 && (value < 1 || value > kTilePts || (value & (value - 1)) != 0))
            return fail(h, COREG_EINVAL, "tile_w must be 0 or a power of two <= 1024");
        h->opt_tile_w = value;
    } else if (n == "n_groups") {
        if (value < 0) return fail(h, COREG_EINVAL, "n_groups must be >= 0");
        h->opt_n_groups = value;
    } else if (n == "skew") {
        (void)value;  // accepted for compatibility: the LDS row skew was measured to lose and is gone
    } else if (n == "shard_world") {
        if (value < 1 || value > 64) return fail(h, COREG_EINVAL, "shard_world must be in [1, 64]");
        h->opt_shard_world = value;
        if (h->opt_shard_rank >= value) h->opt_shard_rank = 0;
    } else if (n == "combo_begin") {
        if (value < 0) return fail(h, COREG_EINVAL, "combo_begin must be >= 0");
        h->opt_combo_begin = value;
    } else if (n == "combo_end") {
        if (value < 0) return fail(h, COREG_EINVAL, "combo_end must be >= 0");
        h->opt_combo_end = value;
    } else if (n == "shard_rank") {
        if (value < 0 || value >= h->opt_shard_world) return fail(h, COREG_EINVAL, "shard_rank must be in [0, shard_world)");
        h->opt_shard_rank = value;
    } else if (n == "border_fix") {
        h->opt_border_fix = value ? 1 : 0;  // 0: the zero lag keeps every border pixel (exact identity map)
    } else if (n == "tap_fix") {
        h->opt_tap_fix = value ? 1 : 0;
    } else if (n == "tap_cap") {
        if (value < 1 || value > (1 << 26)) return fail(h, COREG_EINVAL, "tap_cap must be in [1, 2^26]");
        h->opt_tap_cap = value;
    } else if (n == "pitch") {
        h->opt_pitch = value;  // -1: automatic compile-time window pitch, 0: per-visit pitch, else one of pick_pitch's
    } else if (n == "taper_frac") {
        if (value < -1 || value > 1000) return fail(h, COREG_EINVAL, "taper_frac must be in [-1, 1000]");
        h->opt_taper_frac = value;  // -1: automatic (512 for launches of many rounds, else 0 = equal shares)
    } else if (n == "taper_min") {
        if (value < 16 || value > 1024) return fail(h, COREG_EINVAL, "taper_min must be in [16, 1024]");
        h->opt_taper_min = value;
    } else if (n == "taper_rounds") {
        if (value < 1) return fail(h, COREG_EINVAL, "taper_rounds must be >= 1");
        h->opt_taper_rounds = value;
    } else if (n == "crop_reference") {
        h->opt_crop_reference = value ? 1 : 0;  // 0: the reference preparation uploads the whole source image
    } else if (n == "tile_skip") {
        h->opt_tile_skip = value ? 1 : 0;  // 0: k_precompute evaluates every grid point (tests compare both)
    } else if (n == "h_series") {
        h->opt_h_series = value ? 1 : 0;
    } else if (n == "h_incr") {  // homography sweeps, order 2: advance the affine terms along runs of a grid row
        h->opt_h_incr = value ? 1 : 0;
    } else if (n == "patch_w") {
        if (value < 0 || value > kBlock) return fail(h, COREG_EINVAL, "patch_w must be in [0, 256]");
        h->opt_patch_w = value;
    } else if (n == "lds_bytes") {
        // 160 KiB per CU minus the kernel's static LDS
        if (value < 1024 || value > 159 * 1024) return fail(h, COREG_EINVAL, "lds_bytes must be in [1 KiB, 159 KiB]");
        h->opt_lds_bytes = value;
    } else {
        return fail(h, COREG_EINVAL, "unknown option: " + n);
    }
    return COREG_OK;
}

}  // extern "C"

#include "abi_images.hpp"        // image to align, reference preparation, resampling
#include "abi_sweeps.hpp"        // the sweeps, grid-shared sums, stats
#include "abi_host.hpp"          // host-only helpers


#include "multi.hpp"
