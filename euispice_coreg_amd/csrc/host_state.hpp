// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): device / page-locked buffers, the per-launch fix arguments and the handle (coreg_handle): every field the ABI functions share.
#pragma once
namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            if (e != hipSuccess) return e;
            p = nullptr;
            cap = 0;
        }
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const {
        return (T*)p;
    }
};

struct PinBuf {  // page-locked host staging (async H2D without a host sync)
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// where the pixels of an upload live
enum SrcKind {
    SRC_HOST = 0,    // pageable host memory: staged through the handle's pinned buffers
    SRC_PINNED = 1,  // page-locked host memory every device can DMA from (coreg_multi's shared staging): one async copy
    SRC_DEVICE = 2,  // memory of the handle's GPU: read where it is
};

// what the pixels of an upload are: native float32 / float64, or a FITS data unit's big-endian elements
struct PixFmt {
    int bitpix = 0;  // 0: native pixels (`f32` says which); else the FITS BITPIX of raw big-endian pixels
    bool f32 = false;
    double bscale = 1.0, bzero = 0.0;
    bool raw() const { return bitpix != 0; }
    bool scaled() const { return bscale != 1.0 || bzero != 0.0; }  // (as utils/fits_io.py decides it)
    size_t elem() const { return raw() ? (size_t)(bitpix < 0 ? -bitpix : bitpix) / 8 : (f32 ? 4 : 8); }
    bool swap_only() const { return bitpix == -32 && !scaled(); }  // decoded in place: float32 pixels
    static PixFmt native(bool is_f32) {
        PixFmt f;
        f.f32 = is_f32;
        return f;
    }
};

constexpr int kMaxDevices = 64;
std::mutex g_attr_mutex;

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    bool b_is_next_sweep = false;  // precompute intervals end at the opening event of sweep launch `next_sweep_index`
    size_t next_sweep_index = 0;
};

}  // namespace

// The noise-decided samples of ONE launch as kernel arguments: run once about the global pivots into the
// launch's extra slab and -- when lag-points of the launch are re-evaluated -- a second time about the flagged slots'
// own pivots (launch_sweep, coreg_finalize_sums).  The device lists the arguments point to live until the next sweep.
// (a launch's single-sample lists kept past the next launch of the same sweep: grid-shared plate-carree sweeps)
struct KeptTapLists {
    DevBuf seg_slot, seg_begin, pixel, xw, yw;
    ~KeptTapLists() {
        seg_slot.release();
        seg_begin.release();
        pixel.release();
        xw.release();
        yw.release();
    }
};
struct FixLaunch {
    std::vector<BorderFixArgs> border;
    std::vector<ParityFixArgs> parity;
    TapFixArgs tap = {};
    bool have_tap = false;
    int tap_segs = 0, tap_mode = 0;
    long long tap_count = 0;  // entries of the lists
    std::shared_ptr<KeptTapLists> kept;
    bool small_f32 = true;
    bool empty() const { return border.empty() && parity.empty() && !have_tap; }
};

struct coreg_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // small image
    DevBuf small;
    int sW = 0, sH = 0;
    bool small_f32 = false;
    // reference on grid
    DevBuf ref;
    int gW = 0, gH = 0;
    int ref_dtype = -1;
    // pivots[0] = mean(reference), pivots[1] = mean(small)
    DevBuf pivots, red_sum, red_cnt;
    // geometry tables (+ the host copy they were built from: re-uploaded only when the grid changes)
    DevBuf t_sin_lon, t_cos_lon, t_cos_lat, t_sin_lat;
    CarrTables tabs;
    std::vector<double> tabs_key;
    PinBuf pin_img[2];
    hipEvent_t ev_img[2] = {nullptr, nullptr};
    int pin_img_next = 0;
    // precompute outputs
    DevBuf pts, tile_count, tile_list, tile_cum, group_first, tile_info, tile_bbox;
    DevBuf rf_fix_slab;  // [kNumSums][n_slots]: a launch's noise-decided samples about the flagged slots' own pivots
    DevBuf counters;  // [0]: lag-points re-evaluated by k_finalize during the sweep in flight (reset by its prologue)
    // sweep
    DevBuf lane_params, out_index, partials, out_dev, tmp_img;
    DevBuf up_f64, up_flag;  // upload staging on the device (float64 copy, exactness flag)
    DevBuf up_raw;           // raw FITS elements awaiting their decode (BITPIX other than an unscaled -32)
    DevBuf rice_blob, rice_rand, dec_img;  // tile-compressed images: heap + tile tables, cfitsio's random sequence, a
                                           // decoded reference image (the image to align is decoded in place)
    PrologueArgs pending_prologue = {};  // set by upload_plan, consumed by the sweep's first k_precompute launch
    DevBuf bbox_buf;         // reference_crop: partial bounding boxes
    hipStream_t aux_stream = nullptr;  // side stream of reference_crop (created on first use)
    // The image to align goes up on a stream of its own (round 5): the preparation of the reference image -- bounding
    // box, crop upload, resample -- depends on headers and the reference image only and no longer queues behind the
    // 16 MiB of the image to align; the first call that reads the image (or its pivot) joins the two streams with an event
    // (bind_device).  float32 / byte-swap-only uploads from host memory; everything else stays on `stream`.
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_small = nullptr, ev_main = nullptr;
    bool small_pending = false;
    DevBuf red_sum_up, red_cnt_up;  // device_mean's scratch on up_stream
    int64_t opt_overlap_upload = 1;
    int64_t opt_tap_nan_filter = 2;  // odd orders: list only the near-integer samples that can change the result (k_tap_scan);
                                     // 1: a non-finite pixel anywhere in the union of the footprints, 2: + the sharper
                                     // end-line test where one axis only is near an integer, 0: list them all
    // "async_upload" (opt-in: the caller's image buffer must stay valid and unchanged until the next call that reads the
    // image returns): the staging copies + DMA of coreg_set_small_f32 / _fits run on a worker thread of the handle, so that
    // the calling thread goes on to prepare the reference and plan the sweep meanwhile; joined before the first kernel
    // that reads the image (join_small).  Staging and events of its own: nothing is shared with the calling thread.
    int64_t opt_async_upload = 0;
    std::thread up_thread;
    std::mutex up_m;
    std::condition_variable up_cv;
    std::function<hipError_t()> up_job;
    bool up_has = false, up_stop = false, up_busy = false;
    hipError_t up_rc = hipSuccess;
    PinBuf pin_small[2];
    hipEvent_t ev_pin_small[2] = {nullptr, nullptr};
    int pin_small_next = 0;
    // zero-lag border decision of the helioprojective sub-map path (geometry.hpp WcslibTan): grid pixels the
    // reference's wcslib round trip drops, cached per header
    std::map<std::vector<double>, std::vector<int>> border_cache;
    std::map<std::vector<double>, std::vector<unsigned char>> flags_cache;
    DevBuf border_flags, fix_partial;
    DevBuf border_dev;
    PinBuf pin_border;
    int64_t opt_border_fix = 1;
    // odd spline orders, general case: samples whose coordinate comes back within opt_tap_tol of an integer are
    // re-evaluated with wcslib's own arithmetic (k_tap_scan / k_tap_fix)
    int64_t opt_tap_fix = 1, opt_tap_cap = 1 << 24;
    DevBuf tap_count, tap_segq, tap_list, tap_skip, tap_seg_slot, tap_seg_begin, tap_pixel, tap_xw, tap_yw;
    long long tap_last[3] = {0, 0, 0};  // last sweep: samples listed, lag-points concerned, 1 = list overflowed (no fix)
    // multi-GPU point sharding (coreg_set_option "shard_world" / "shard_rank"): a sweep covers this rank's share of the
    // tile groups and leaves the six sums per lag slot in `sums`; coreg_finalize_sums turns the all-reduced sums into
    // coefficients
    int64_t opt_shard_world = 1, opt_shard_rank = 0;
    // multi-GPU combination sharding ("combo_begin" / "combo_end"): the NEXT sweep covers only the (cdelt1, cdelt2, crota)
    // combinations [begin, end) of the lag set's inner C-order index; consumed (reset to "all") by that sweep
    int64_t opt_combo_begin = 0, opt_combo_end = 0;
    DevBuf sums;
    long long sums_slots = 0;  // slots of the pending sharded sweep (all its launches)
    struct PendingFinalize {
        long long slot_off, n_slots, lag_begin;
        const long long* outidx_dev;
        int residus;
        // what coreg_finalize_sums needs to re-evaluate the ill-conditioned lag-points of this launch once the ranks' sums
        // are added (the flags come from the REDUCED sums): the launch's refine arguments and, when later launches of the
        // same sweep have overwritten the compacted points, how to compute them again
        RefineArgs refine;
        std::function<int(coreg_handle*)> replay_precompute;
        FixLaunch fixes;  // the launch's noise-decided samples, for the second run about the flagged slots' pivots
    };
    std::function<int(coreg_handle*)> last_precompute;  // the precompute launch the next launch_sweep follows
    DevBuf rf_flags, rf_pivots, rf_list, rf_head, rf_partial;  // work space of the re-evaluation (kernels.hpp: RefineArgs)
    std::vector<PendingFinalize> pending_fin;
    DevBuf fin_outidx;        // copy of the output indices of the pending sharded sweep
    long long pending_n_out = 0;

    // options
    int64_t opt_crop_reference = 1;
    int64_t opt_taper_min = 128, opt_taper_frac = -1, opt_taper_rounds = 6;  // tapered group shares (pick_taper)
    int64_t opt_use_lds = 1, opt_clean_path = 1, opt_refine = 1, opt_refine_cond_log10 = 5, opt_tile_w = 0, opt_n_groups = 0, opt_lds_bytes = (159 * 1024 * kPointGroups) / 4, opt_patch_w = 0, opt_h_series = 1, opt_h_incr = 1, opt_tile_skip = 1, opt_pitch = -1;

    coreg_stats stats;
    bool stats_pending = false;   // a device-output sweep is in flight: timings are collected on demand
    PinBuf pin_info;              // tile_info read-back of the in-flight sweep
    std::vector<EventPair> ev_sweep, ev_pre;
    size_t ev_sweep_used = 0, ev_pre_used = 0;
    hipEvent_t ev_t1 = nullptr;
    // The plan staging (lag parameters in pinned memory) is double-buffered: slot k is rewritten only when the sweep
    // that last used it has ended (its end event), so the host can plan sweep n + 1 while the GPU runs sweep n, with no
    // extra event between the kernels.  ev_t1 is an alias of the current slot's end event.
    hipEvent_t ev_end[2] = {nullptr, nullptr};
    PinBuf pin_plan[2];
    int plan_slot = 0;
    bool plan_open = false;  // upload_plan has staged a plan that no end_sweep has closed yet (a sweep that failed midway)
};

