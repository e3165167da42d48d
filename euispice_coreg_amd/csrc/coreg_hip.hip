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

namespace {

int fail(coreg_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(h, COREG_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

#define RETCHK(expr)             \
    do {                         \
        int _r = (expr);         \
        if (_r != COREG_OK) return _r; \
    } while (0)

// COREG_TRACE=1: host-side timestamps (microseconds since the first one) of the hand-over's stages on stderr
inline void trace(const char* what) {
    static const bool on = [] {
        const char* e = std::getenv("COREG_TRACE");
        return e && std::atoi(e) == 1;
    }();
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "[coreg %9.1f us] %s\n", us, what);
}
int bind_device_nowait(coreg_handle* h) {
    HIPCHK(hipSetDevice(h->device));
    return COREG_OK;
}
// every entry point but the reference preparation: work enqueued on the handle's stream from here on sees the image to
// align a preceding set_small put on the upload stream
// the image to align is on its way on the upload stream (possibly still being issued by the handle's upload thread):
// make the handle's stream wait for it
int join_small(coreg_handle* h) {
    if (!h->small_pending) return COREG_OK;
    trace("join_small: waiting for the upload thread");
    hipError_t worker_rc = hipSuccess;
    {
        std::unique_lock<std::mutex> lk(h->up_m);
        h->up_cv.wait(lk, [&] { return !h->up_busy && !h->up_has; });
        worker_rc = h->up_rc;
        h->up_rc = hipSuccess;
    }
    h->small_pending = false;
    if (worker_rc != hipSuccess)
        return fail(h, COREG_EHIP, std::string("asynchronous upload of the image to align: ") + hipGetErrorString(worker_rc));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_small, 0));
    trace("join_small: joined");
    return COREG_OK;
}
int bind_device(coreg_handle* h) {
    HIPCHK(hipSetDevice(h->device));
    return join_small(h);
}
void upload_thread_main(coreg_handle* h) {
    for (;;) {
        std::function<hipError_t()> job;
        {
            std::unique_lock<std::mutex> lk(h->up_m);
            h->up_cv.wait(lk, [&] { return h->up_stop || h->up_has; });
            if (h->up_stop) return;
            job = std::move(h->up_job);
            h->up_has = false;
            h->up_busy = true;
        }
        const hipError_t rc = job();
        {
            std::lock_guard<std::mutex> lk(h->up_m);
            h->up_rc = rc;
            h->up_busy = false;
        }
        h->up_cv.notify_all();
    }
}
void post_upload(coreg_handle* h, std::function<hipError_t()> job) {
    if (!h->up_thread.joinable()) h->up_thread = std::thread(upload_thread_main, h);
    {
        std::lock_guard<std::mutex> lk(h->up_m);
        h->up_job = std::move(job);
        h->up_has = true;
    }
    h->up_cv.notify_all();
}
// the stream an upload of the image to align runs on: the upload stream, made to wait for what the handle's stream has
// been given so far (an earlier sweep may still be reading the old image), or the handle's stream itself
int begin_small_upload(coreg_handle* h, hipStream_t* s) {
    *s = h->stream;
    if (!h->opt_overlap_upload) return COREG_OK;
    if (!h->up_stream) {
        HIPCHK(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->ev_small, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(h->ev_main, h->stream));
    HIPCHK(hipStreamWaitEvent(h->up_stream, h->ev_main, 0));
    *s = h->up_stream;
    return COREG_OK;
}
int end_small_upload(coreg_handle* h, hipStream_t s) {
    if (s == h->stream) return COREG_OK;
    HIPCHK(hipEventRecord(h->ev_small, s));
    h->small_pending = true;
    return COREG_OK;
}

EventPair* next_event(coreg_handle* h, std::vector<EventPair>& v, size_t& used) {
    if (used == v.size()) {
        EventPair e;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return nullptr;
        v.push_back(e);
    }
    return &v[used++];
}

template <typename T>
int device_mean(coreg_handle* h, const T* v, long long n, double* mean_dev, hipStream_t s = nullptr) {
    const int nb = 256;
    const bool up = s && s != h->stream;  // (the upload stream has scratch of its own)
    if (!s) s = h->stream;
    DevBuf& sum = up ? h->red_sum_up : h->red_sum;
    DevBuf& cnt = up ? h->red_cnt_up : h->red_cnt;
    HIPCHK(sum.reserve(nb * sizeof(double)));
    HIPCHK(cnt.reserve(nb * sizeof(long long)));
    hipLaunchKernelGGL((k_sum_finite<T>), dim3(nb), dim3(256), 0, s, v, n, sum.as<double>(), cnt.as<long long>());
    hipLaunchKernelGGL(k_mean_final, dim3(1), dim3(64), 0, s, sum.as<double>(), cnt.as<long long>(), nb, mean_dev);
    HIPCHK(hipGetLastError());
    return COREG_OK;
}

// host -> device through pinned staging: worker threads fill the staging buffer segment by segment while the DMA
// engine drains the previous segment (a plain hipMemcpy from pageable memory runs at a fraction of the link rate).
// The workers are a small persistent pool (creating threads per segment costs as much as the copy itself).
class CopyPool {
public:
    static CopyPool& get() {
        static CopyPool p;
        return p;
    }
    void copy(void* dst, const void* src, size_t bytes) {
        static const size_t min_per_thread = [] {
            const char* e = std::getenv("COREG_UPLOAD_MIN_KIB");
            const int v = e ? std::atoi(e) : 0;
            return (size_t)(v > 0 ? v : 512) << 10;
        }();
        const unsigned nt = (unsigned)std::min<size_t>(workers_.size() + 1, std::max<size_t>(1, bytes / min_per_thread));
        if (nt <= 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        std::lock_guard<std::mutex> use(use_);  // one parallel copy at a time
        const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
        {
            std::lock_guard<std::mutex> lk(m_);
            dst_ = (char*)dst;
            src_ = (const char*)src;
            bytes_ = bytes;
            per_ = per;
            rows_ = 0;
            n_parts_ = nt;
            next_ = 1;  // part 0 is the caller's
            pending_ = nt - 1;
            ++epoch_;
        }
        cv_.notify_all();
        copy_stream((char*)dst, (const char*)src, std::min(per, bytes));
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }
    // `rows` rows of `row_bytes` bytes, `src_pitch` bytes apart in the source, packed contiguously into dst
    void copy_rows(void* dst, const void* src, size_t rows, size_t row_bytes, size_t src_pitch) {
        const size_t min_per_thread = (size_t)512 << 10;
        const unsigned nt = (unsigned)std::min<size_t>(
            std::min<size_t>(workers_.size() + 1, std::max<size_t>(1, rows)), std::max<size_t>(1, rows * row_bytes / min_per_thread));
        if (nt <= 1) {
            for (size_t r = 0; r < rows; ++r)
                std::memcpy((char*)dst + r * row_bytes, (const char*)src + r * src_pitch, row_bytes);
            return;
        }
        std::lock_guard<std::mutex> use(use_);
        {
            std::lock_guard<std::mutex> lk(m_);
            dst_ = (char*)dst;
            src_ = (const char*)src;
            rows_ = rows;
            row_bytes_ = row_bytes;
            src_pitch_ = src_pitch;
            per_ = (rows + nt - 1) / nt;  // rows per part
            n_parts_ = nt;
            next_ = 1;
            pending_ = nt - 1;
            ++epoch_;
        }
        cv_.notify_all();
        part(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    CopyPool() {
        const char* e = std::getenv("COREG_UPLOAD_THREADS");
        int want = e ? std::atoi(e) : 0;
        if (want <= 0) want = 12;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw > 0) want = std::min<int>(want, (int)hw);
        for (int i = 1; i < std::min(want, 64); ++i) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void run() {
        unsigned long long seen = 0;
        for (;;) {
            unsigned idx;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && next_ < n_parts_); });
                if (stop_) return;
                idx = next_++;
                if (next_ >= n_parts_) seen = epoch_;
            }
            part(idx);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    // The destination is page-locked staging the CPU never reads back: non-temporal stores spare the read-for-ownership
    // of every destination line (glibc's memcpy only switches to them far above the 100-500 KB a worker copies).
    // COREG_UPLOAD_NT=0 keeps memcpy.
    static void copy_stream(char* dst, const char* src, size_t n) {
        typedef long long v4 __attribute__((vector_size(32), aligned(32)));
        static const bool nt = [] {
            const char* e = std::getenv("COREG_UPLOAD_NT");
            return !(e && std::atoi(e) == 0);
        }();
        if (!nt || n < 4096) {
            std::memcpy(dst, src, n);
            return;
        }
        const size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
        if (head) std::memcpy(dst, src, head);
        size_t i = head;
        for (; i + 128 <= n; i += 128) {
            v4 a, b, c, d;
            std::memcpy(&a, src + i, 32);
            std::memcpy(&b, src + i + 32, 32);
            std::memcpy(&c, src + i + 64, 32);
            std::memcpy(&d, src + i + 96, 32);
            __builtin_nontemporal_store(a, (v4*)(dst + i));
            __builtin_nontemporal_store(b, (v4*)(dst + i + 32));
            __builtin_nontemporal_store(c, (v4*)(dst + i + 64));
            __builtin_nontemporal_store(d, (v4*)(dst + i + 96));
        }
        if (i < n) std::memcpy(dst + i, src + i, n - i);
        std::atomic_thread_fence(std::memory_order_seq_cst);  // the DMA that follows must see the streamed lines
    }
    void part(unsigned p) {
        if (rows_ > 0) {
            const size_t lo = std::min(rows_, (size_t)p * per_), hi = std::min(rows_, lo + per_);
            for (size_t r = lo; r < hi; ++r) std::memcpy(dst_ + r * row_bytes_, src_ + r * src_pitch_, row_bytes_);
        } else {
            const size_t lo = std::min(bytes_, (size_t)p * per_), hi = std::min(bytes_, lo + per_);
            if (hi > lo) copy_stream(dst_ + lo, src_ + lo, hi - lo);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, use_;
    std::condition_variable cv_, done_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0, per_ = 0, rows_ = 0, row_bytes_ = 0, src_pitch_ = 0;
    unsigned n_parts_ = 0, next_ = 0, pending_ = 0;
    unsigned long long epoch_ = 0;
    bool stop_ = false;
};
void parallel_memcpy(void* dst, const void* src, size_t bytes) { CopyPool::get().copy(dst, src, bytes); }
void parallel_copy_rows(void* dst, const void* src, size_t rows, size_t row_bytes, size_t src_pitch) {
    CopyPool::get().copy_rows(dst, src, rows, row_bytes, src_pitch);
}

int staged_upload(coreg_handle* h, void* dev, const void* host, size_t bytes, hipStream_t stream = nullptr) {
    if (!stream) stream = h->stream;
    // two staging buffers used alternately, each guarded by an event recorded behind its last copy: filling the
    // buffer for this upload overlaps the DMA (and whatever else the stream is doing) of the previous one
    const int k = h->pin_img_next;
    h->pin_img_next ^= 1;
    if (!h->ev_img[k]) HIPCHK(hipEventCreateWithFlags(&h->ev_img[k], hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(h->ev_img[k]));  // the upload that last used this buffer has left it
    HIPCHK(h->pin_img[k].reserve(bytes));
    char* pin = (char*)h->pin_img[k].p;
    // segments: small at first so that the DMA engine starts early, then larger
    static const size_t seg_max = [] {
        const char* e = std::getenv("COREG_UPLOAD_SEGMENT_MIB");
        const int v = e ? std::atoi(e) : 0;
        return (size_t)(v > 0 ? v : 6) << 20;
    }();
    size_t seg = (size_t)2 << 20;
    for (size_t off = 0; off < bytes; off += seg, seg = std::min(seg * 2, seg_max)) {
        const size_t len = std::min(seg, bytes - off);
        parallel_memcpy(pin + off, (const char*)host + off, len);
        HIPCHK(hipMemcpyAsync((char*)dev + off, pin + off, len, hipMemcpyHostToDevice, stream));
    }
    HIPCHK(hipEventRecord(h->ev_img[k], stream));
    return COREG_OK;
}

// the same on the handle's upload thread: staging and events of its own, plain HIP error codes (h->err belongs to the
// calling thread), then the byte swap of a BITPIX = -32 data unit and the pivot of the image, all on stream `s`
hipError_t upload_small_worker(coreg_handle* h, void* dev, const void* host, size_t n_elem, bool swap32, hipStream_t s) {
    trace("worker: upload begins");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return e;
    const size_t bytes = n_elem * 4;
    const int k = h->pin_small_next;
    h->pin_small_next ^= 1;
    if (!h->ev_pin_small[k]) e = hipEventCreateWithFlags(&h->ev_pin_small[k], hipEventDisableTiming);
    else e = hipEventSynchronize(h->ev_pin_small[k]);
    if (e != hipSuccess) return e;
    if ((e = h->pin_small[k].reserve(bytes)) != hipSuccess) return e;
    char* pin = (char*)h->pin_small[k].p;
    size_t seg = (size_t)2 << 20;
    for (size_t off = 0; off < bytes; off += seg, seg = std::min(seg * 2, (size_t)6 << 20)) {
        const size_t len = std::min(seg, bytes - off);
        parallel_memcpy(pin + off, (const char*)host + off, len);
        if ((e = hipMemcpyAsync((char*)dev + off, pin + off, len, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    }
    if ((e = hipEventRecord(h->ev_pin_small[k], s)) != hipSuccess) return e;
    const int nb = (int)std::min<size_t>((n_elem + 255) / 256, 4096);
    if (swap32) hipLaunchKernelGGL(k_fits_swap32, dim3(nb), dim3(256), 0, s, (unsigned int*)dev, (long long)n_elem);
    if ((e = h->red_sum_up.reserve(256 * sizeof(double))) != hipSuccess) return e;
    if ((e = h->red_cnt_up.reserve(256 * sizeof(long long))) != hipSuccess) return e;
    hipLaunchKernelGGL((k_sum_finite<float>), dim3(256), dim3(256), 0, s, (const float*)dev, (long long)n_elem,
                       h->red_sum_up.as<double>(), h->red_cnt_up.as<long long>());
    hipLaunchKernelGGL(k_mean_final, dim3(1), dim3(64), 0, s, h->red_sum_up.as<double>(), h->red_cnt_up.as<long long>(), 256,
                       h->pivots.as<double>() + 1);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = hipEventRecord(h->ev_small, s);
    trace("worker: upload issued");
    return e;
}

// A float64 image (host: staged upload; device: the caller's buffer) is kept as float32 on the device when every finite
// value is exactly representable (FITS BITPIX=-32 / integer data cast to float64), else as float64.  The test and the
// conversion run on the GPU.  src_on_device: `img` is device memory, read by work enqueued on the handle's stream.
int upload_image(coreg_handle* h, const double* img, size_t n, DevBuf& buf, bool* is_f32, SrcKind kind = SRC_HOST) {
    const bool src_on_device = kind == SRC_DEVICE;
    HIPCHK(h->up_flag.reserve(sizeof(int)));
    const double* src = img;
    if (!src_on_device) {
        HIPCHK(h->up_f64.reserve(n * sizeof(double)));
        if (kind == SRC_PINNED)
            HIPCHK(hipMemcpyAsync(h->up_f64.p, img, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        else
            RETCHK(staged_upload(h, h->up_f64.p, img, n * sizeof(double)));
        src = h->up_f64.as<double>();
    }
    HIPCHK(hipMemsetAsync(h->up_flag.p, 0, sizeof(int), h->stream));
    const int nb = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_f32_exact, dim3(nb), dim3(256), 0, h->stream, src, (long long)n, h->up_flag.as<int>());
    HIPCHK(hipGetLastError());
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, h->up_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *is_f32 = flag == 0;
    if (*is_f32) {
        HIPCHK(buf.reserve(n * sizeof(float)));
        hipLaunchKernelGGL(k_f64_to_f32, dim3(nb), dim3(256), 0, h->stream, src, (long long)n, buf.as<float>());
        HIPCHK(hipGetLastError());
    } else if (src_on_device) {
        HIPCHK(buf.reserve(n * sizeof(double)));
        HIPCHK(hipMemcpyAsync(buf.p, src, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    } else {
        std::swap(buf.p, h->up_f64.p);  // the float64 copy becomes the image
        std::swap(buf.cap, h->up_f64.cap);
    }
    return COREG_OK;
}

// raw FITS elements on the device -> the pixels the kernels read.  BITPIX = -32 without scaling: `raw_dev` IS buf.p, the
// byte swap runs in place and the image is float32.  Everything else: float64(stored) * bscale + bzero into up_f64, then
// the same float32-exactness test and conversion a float64 upload gets (upload_image).
int fits_decode(coreg_handle* h, const PixFmt& fmt, void* raw_dev, size_t n, DevBuf& buf, bool* is_f32) {
    const int nb = (int)std::min<size_t>((n + 255) / 256, 4096);
    if (fmt.swap_only()) {
        hipLaunchKernelGGL(k_fits_swap32, dim3(nb), dim3(256), 0, h->stream, (unsigned int*)raw_dev, (long long)n);
        HIPCHK(hipGetLastError());
        *is_f32 = true;
        return COREG_OK;
    }
    HIPCHK(h->up_f64.reserve(n * sizeof(double)));
    hipLaunchKernelGGL(k_fits_to_f64, dim3(nb), dim3(256), 0, h->stream, (const void*)raw_dev, fmt.bitpix,
                       fmt.scaled() ? 1 : 0, fmt.bscale, fmt.bzero, (long long)n, h->up_f64.as<double>());
    HIPCHK(hipGetLastError());
    return upload_image(h, h->up_f64.as<double>(), n, buf, is_f32, SRC_DEVICE);
}

int check_fits(coreg_handle* h, const coreg_fits_pixels* px, PixFmt* fmt) {
    if (!px || !px->data) return fail(h, COREG_EINVAL, "fits pixels: null pointer");
    const int b = px->bitpix;
    if (b != 8 && b != 16 && b != 32 && b != 64 && b != -32 && b != -64)
        return fail(h, COREG_EINVAL, "fits pixels: BITPIX must be 8, 16, 32, 64, -32 or -64");
    if (!std::isfinite(px->bscale) || !std::isfinite(px->bzero))
        return fail(h, COREG_EINVAL, "fits pixels: BSCALE / BZERO not finite");
    fmt->bitpix = b;
    fmt->bscale = px->bscale;
    fmt->bzero = px->bzero;
    return COREG_OK;
}

// ---- tile-compressed FITS images (csrc/ricecomp.hpp) -----------------------------------------------------------------
// One WORKGROUP (one wave) per tile.  A tile's bit stream is sequential, so ONE lane decodes it -- alone in its wave,
// i.e. without the divergence 64 independent streams per wave would serialise -- between two parallel phases: all 64
// lanes stage the tile's compressed bytes in LDS (coalesced loads; the decoder then reads LDS, not one global byte per
// dependent load), lane 0 leaves the decoded integers in LDS, and all 64 lanes turn them into pixel values (scale,
// zero, dither, NaN) and store them row by row, coalesced.  Tiles too large for the buffers take the direct path.
// The two buffers are sized per launch (dynamic LDS: q_cap integers, then stream_cap bytes) from the image's largest tile
// and longest stream, up to the limits below: a 2048-pixel row of an EUI image needs 8 KB + ~3 KB, so every tile of the
// image is resident at once (16 waves per CU decoding) instead of two rounds of five.
constexpr int kRiceStream = 16 * 1024;  // most bytes of compressed stream staged (a 4096-pixel row of verbatim 4-byte values)
constexpr int kRicePixels = 4096;       // most decoded integers buffered
__global__ void __launch_bounds__(64) k_rice_tiles(const coregrice::TileImage t, int* status, int q_cap, int stream_cap) {
    extern __shared__ int32_t rice_lds[];
    int32_t* const qbuf = rice_lds;
    unsigned char* const stream = (unsigned char*)(rice_lds + q_cap);
    const int n = blockIdx.x;
    const long long off = t.tile_offset[n];
    const int len = t.tile_nbytes[n];
    const coregrice::TileBox box = coregrice::tile_box(t, n);
    const int npx = box.tw * box.th;
    const bool in_heap = len > 0 && off >= 0 && off + len <= t.heap_bytes;
    const bool staged = in_heap && len <= stream_cap && npx <= q_cap;
    if (!staged) {
        if (threadIdx.x == 0) {
            const int e = coregrice::decode_tile(t, n);
            if (e) atomicOr(status, e);
        }
        return;
    }
    {
        const unsigned char* src = t.heap + off;
        for (int i = threadIdx.x; i < len; i += 64) stream[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        coregrice::QSink sink = {qbuf, 0};
        const int e = coregrice::rice_decode_tile(stream, len, npx, t.blocksize, t.bytepix, sink);
        if (e) atomicOr(status, 1);
    }
    __syncthreads();
    const double scale = t.zscale ? t.zscale[n] : t.zscale0, zero = t.zzero ? t.zzero[n] : t.zzero0;
    const bool dith = t.quantize == coregrice::Q_DITHER_1 || t.quantize == coregrice::Q_DITHER_2;
    const int iseed = dith ? coregrice::dither_seed(t, n) : 0;
    for (int i = threadIdx.x; i < npx; i += 64) {
        const int ty = i / box.tw, tx = i - ty * box.tw;
        const double v = coregrice::pixel_value(t, scale, zero, qbuf[i], dith ? coregrice::dither_index(t.randoms, iseed, i) : 0);
        const long long at = (long long)(box.y0 + ty) * t.naxis1 + (box.x0 + tx);
        if (t.out_dtype == coregrice::OUT_F32) ((float*)t.out)[at] = (float)v;
        else ((double*)t.out)[at] = v;
    }
}

const char* check_tiled(const coreg_fits_tiled* t) {
    if (!t || !t->heap || !t->tile_offset || !t->tile_nbytes) return "tiled image: null pointer";
    if (t->heap_bytes < 1 || t->n_tiles < 1) return "tiled image: empty heap or no tiles";
    if (t->naxis1 < 1 || t->naxis2 < 1 || t->ztile1 < 1 || t->ztile2 < 1) return "tiled image: bad image / tile shape";
    const long long ntx = (t->naxis1 + t->ztile1 - 1) / t->ztile1, nty = (t->naxis2 + t->ztile2 - 1) / t->ztile2;
    if (ntx * nty != t->n_tiles) return "tiled image: n_tiles does not match the tiling";
    if (t->bytepix != 1 && t->bytepix != 2 && t->bytepix != 4) return "tiled image: BYTEPIX must be 1, 2 or 4";
    if (t->blocksize < 1 || t->blocksize > 1024) return "tiled image: bad BLOCKSIZE";
    const int b = t->zbitpix;
    if (b != 8 && b != 16 && b != 32 && b != -32 && b != -64) return "tiled image: ZBITPIX must be 8, 16, 32, -32 or -64";
    if (t->quantize < 0 || t->quantize > 3) return "tiled image: bad quantize method";
    if ((b < 0) != (t->quantize != 0)) return "tiled image: quantize method and ZBITPIX disagree";
    if (b < 0 && t->bytepix != 4) return "tiled image: quantized floats are 4-byte integers";
    if (b > 0 && t->bytepix * 8 != b) return "tiled image: BYTEPIX and ZBITPIX disagree";
    if ((t->zscale == nullptr) != (t->zzero == nullptr)) return "tiled image: ZSCALE / ZZERO must come together";
    return nullptr;
}

void fill_tile_image(const coreg_fits_tiled& t, coregrice::TileImage* im) {
    std::memset(im, 0, sizeof(*im));
    im->naxis1 = t.naxis1;
    im->naxis2 = t.naxis2;
    im->ztile1 = t.ztile1;
    im->ztile2 = t.ztile2;
    im->bytepix = t.bytepix;
    im->blocksize = t.blocksize;
    im->zbitpix = t.zbitpix;
    im->quantize = t.quantize;
    im->dither0 = t.dither0;
    im->has_blank = t.has_blank || t.zbitpix < 0;
    im->blank = t.has_blank ? t.blank : coregrice::kNullValue;
    im->scaled = (t.bscale != 1.0 || t.bzero != 0.0) ? 1 : 0;
    im->bscale = t.bscale;
    im->bzero = t.bzero;
    im->zscale0 = t.zscale0;
    im->zzero0 = t.zzero0;
    im->heap_bytes = t.heap_bytes;
    im->n_tiles = t.n_tiles;
}

// compressed bytes + tile tables up, one thread per tile decodes into `pix` (float32 for ZBITPIX = -32, else the float64
// pixels go through the float32-exactness test of every float64 upload)
int decode_tiled_device(coreg_handle* h, const coreg_fits_tiled* t, DevBuf& pix, bool* is_f32) {
    if (const char* why = check_tiled(t)) return fail(h, COREG_EINVAL, why);
    if (!std::isfinite(t->bscale) || !std::isfinite(t->bzero)) return fail(h, COREG_EINVAL, "tiled image: BSCALE / BZERO");
    const size_t n = (size_t)t->naxis1 * t->naxis2, nt = (size_t)t->n_tiles;
    for (size_t k = 0; k < nt; ++k)
        if (t->tile_nbytes[k] <= 0)
            return fail(h, COREG_ENOTIMPL, "tiled image: a tile is not Rice-coded (decode it on the host: "
                                           "coreg_decode_tiled_host + the GZIP_COMPRESSED_DATA column)");
    // blob layout: [heap][pad][tile_offset: int64 x nt][zscale: f64 x nt][zzero: f64 x nt][tile_nbytes: int32 x nt]
    const size_t heap_pad = ((size_t)t->heap_bytes + 15) & ~(size_t)15;
    const bool per_tile = t->zscale != nullptr;
    const size_t tbl_bytes = nt * 8 + (per_tile ? nt * 16 : 0) + nt * 4;
    HIPCHK(h->rice_blob.reserve(heap_pad + tbl_bytes));
    char* blob = h->rice_blob.as<char>();
    RETCHK(staged_upload(h, blob, t->heap, (size_t)t->heap_bytes));
    std::vector<char> tbl(tbl_bytes);
    size_t at = 0;
    std::memcpy(tbl.data() + at, t->tile_offset, nt * 8);
    at += nt * 8;
    if (per_tile) {
        std::memcpy(tbl.data() + at, t->zscale, nt * 8);
        at += nt * 8;
        std::memcpy(tbl.data() + at, t->zzero, nt * 8);
        at += nt * 8;
    }
    std::memcpy(tbl.data() + at, t->tile_nbytes, nt * 4);
    RETCHK(staged_upload(h, blob + heap_pad, tbl.data(), tbl_bytes));  // (copied into pinned staging before returning)
    if (!h->rice_rand.p) {
        std::vector<float> r(coregrice::kNRandom);
        coregrice::init_randoms(r.data());
        HIPCHK(h->rice_rand.reserve(r.size() * sizeof(float)));
        HIPCHK(hipMemcpy(h->rice_rand.p, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    coregrice::TileImage im;
    fill_tile_image(*t, &im);
    im.heap = (const unsigned char*)blob;
    at = heap_pad;
    im.tile_offset = (const int64_t*)(blob + at);
    at += nt * 8;
    if (per_tile) {
        im.zscale = (const double*)(blob + at);
        at += nt * 8;
        im.zzero = (const double*)(blob + at);
        at += nt * 8;
    }
    im.tile_nbytes = (const int32_t*)(blob + at);
    im.randoms = h->rice_rand.as<float>();
    const bool direct_f32 = t->zbitpix == -32;
    if (direct_f32) {
        HIPCHK(pix.reserve(n * sizeof(float)));
        im.out = pix.p;
        im.out_dtype = coregrice::OUT_F32;
    } else {
        HIPCHK(h->up_f64.reserve(n * sizeof(double)));
        im.out = h->up_f64.p;
        im.out_dtype = coregrice::OUT_F64;
    }
    HIPCHK(h->up_flag.reserve(sizeof(int)));
    HIPCHK(hipMemsetAsync(h->up_flag.p, 0, sizeof(int), h->stream));
    int max_len = 0;
    for (size_t k = 0; k < nt; ++k) max_len = std::max(max_len, (int)t->tile_nbytes[k]);
    const int q_cap = (int)std::min<long long>((long long)std::min(t->ztile1, t->naxis1) * std::min(t->ztile2, t->naxis2), kRicePixels);
    const int stream_cap = std::min((max_len + 15) & ~15, kRiceStream);
    hipLaunchKernelGGL(k_rice_tiles, dim3((unsigned)nt), dim3(64), (size_t)q_cap * 4 + stream_cap, h->stream, im,
                       h->up_flag.as<int>(), q_cap, stream_cap);
    HIPCHK(hipGetLastError());
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, h->up_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (flag) return fail(h, COREG_EINVAL, "tiled image: a Rice stream is truncated or corrupt");
    if (direct_f32) {
        *is_f32 = true;
        return COREG_OK;
    }
    return upload_image(h, h->up_f64.as<double>(), n, pix, is_f32, SRC_DEVICE);
}

int upload_carr_tables(coreg_handle* h, const coreg_carr_grid& g, const coreg_wcs2d& hdr, CarrDev* dev) {
    if (g.n_lon < 1 || g.n_lat < 1) return fail(h, COREG_EINVAL, "carrington grid: n_lon/n_lat must be >= 1");
    // key: everything the tables depend on (caller-supplied latitude trig by value)
    std::vector<double> key = {g.lon0, g.lon1, (double)g.n_lon, g.lat0, g.lat1, (double)g.n_lat, hdr.crln_obs,
                               g.lat_cos ? 1.0 : 0.0, g.lat_sin ? 1.0 : 0.0};
    if (g.lat_cos) key.insert(key.end(), g.lat_cos, g.lat_cos + g.n_lat);
    if (g.lat_sin) key.insert(key.end(), g.lat_sin, g.lat_sin + g.n_lat);
    if (key != h->tabs_key || !h->t_sin_lon.p) {
        CarrTables& t = h->tabs;
        carr_tables(g, hdr.crln_obs, t);
        HIPCHK(h->t_sin_lon.reserve(g.n_lon * sizeof(double)));
        HIPCHK(h->t_cos_lon.reserve(g.n_lon * sizeof(double)));
        HIPCHK(h->t_cos_lat.reserve(g.n_lat * sizeof(float)));
        HIPCHK(h->t_sin_lat.reserve(g.n_lat * sizeof(float)));
        // h->tabs outlives the copies (it is only rebuilt after the next key mismatch, behind this same stream)
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpyAsync(h->t_sin_lon.p, t.sin_lon.data(), g.n_lon * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->t_cos_lon.p, t.cos_lon.data(), g.n_lon * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->t_cos_lat.p, t.cos_lat.data(), g.n_lat * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->t_sin_lat.p, t.sin_lat.data(), g.n_lat * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->tabs_key.swap(key);
    }
    dev->sin_lon = h->t_sin_lon.as<double>();
    dev->cos_lon = h->t_cos_lon.as<double>();
    dev->cos_lat = h->t_cos_lat.as<float>();
    dev->sin_lat = h->t_sin_lat.as<float>();
    dev->n_lon = g.n_lon;
    dev->n_lat = g.n_lat;
    return COREG_OK;
}

void set_carr_common(CarrDev* dev, const CarrCommon& c) {
    dev->dist = c.dist;
    dev->cb = c.cb;
    dev->sb = c.sb;
    dev->cr = c.cr;
    dev->sr = c.sr;
    dev->cdelt1 = c.cdelt1;
    dev->cdelt2 = c.cdelt2;
}

// host mirror of kernels.hpp carr_term (tile-shape heuristics only)
bool carr_term_host(const CarrTables& t, const CarrCommon& c, int i, int j, double* t0, double* t1) {
    const double cl = (double)t.cos_lat[j], y = (double)t.sin_lat[j];
    const double x = cl * t.sin_lon[i], z = cl * t.cos_lon[i];
    const double zz = z * c.cb + y * c.sb, yy = y * c.cb - z * c.sb;
    const double yr = yy * c.cr - x * c.sr, xr = x * c.cr + yy * c.sr;
    const double zd = c.dist - zz;
    *t0 = std::atan(xr / zd) * kRad2Deg * 3600.0 / c.cdelt1;
    *t1 = std::atan(yr / zd) * kRad2Deg * 3600.0 / c.cdelt2;
    return zz >= 0.0;
}

template <typename F>
int dispatch_resample(coreg_handle* h, int mode, int order, bool ts_f32, bool out_f32, const ResampleArgs& a, F) {
    const long long n = (long long)a.gw * a.gh;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    ResampleArgs b = a;
    b.order_rt = order;
#define RS(M, O, TS, TO) hipLaunchKernelGGL((k_resample<M, O, TS, TO>), grid, block, 0, h->stream, b)
#define RS_T(M, O)                                  \
    do {                                            \
        if (ts_f32) {                               \
            if (out_f32) RS(M, O, float, float);    \
            else RS(M, O, float, double);           \
        } else {                                    \
            if (out_f32) RS(M, O, double, float);   \
            else RS(M, O, double, double);          \
        }                                           \
    } while (0)
    if (mode == MODE_TRANSLATE) {
        if (order == 2) RS_T(MODE_TRANSLATE, 2);
        else if (order == 1) RS_T(MODE_TRANSLATE, 1);
        else RS_T(MODE_TRANSLATE, ORDER_RT);
    } else if (mode == MODE_CAR) {
        RS_T(MODE_CAR, ORDER_RT);  // once per call, on small maps: the run-time-order gather serves every order
    } else {
        if (order == 2) RS_T(MODE_HOMOGRAPHY, 2);
        else if (order == 1) RS_T(MODE_HOMOGRAPHY, 1);
        else RS_T(MODE_HOMOGRAPHY, ORDER_RT);
    }
#undef RS_T
#undef RS
    HIPCHK(hipGetLastError());
    return COREG_OK;
}

int check_order(coreg_handle* h, int order) {
    // scipy.ndimage.map_coordinates accepts spline orders 0..5 (utils/Util.py:98-102); 1 and 2 run on the tuned
    // kernels, the others on the run-time-order variant
    if (order < 0 || order > 5)
        return fail(h, COREG_EINVAL, "reprojection_order must be in 0..5 (got " + std::to_string(order) + ")");
    return COREG_OK;
}

// ---- lag batching ---------------------------------------------------------------------------------------------
struct LagDims {
    int n1, n2, n3, n4, n5;
    long long nc;  // (cdelt1, cdelt2, crota) combinations this sweep covers: n3*n4*n5, or the "combo_begin/_end" range
    long long c0;  // first of them in the lag set's inner C-order index ((i3 * n4 + i4) * n5 + i5)
    long long total() const { return (long long)n1 * n2 * nc; }
    void inner(long long c, int* i3, int* i4, int* i5) const {  // c in [0, nc)
        const long long g = c + c0;
        *i5 = (int)(g % n5);
        *i4 = (int)((g / n5) % n4);
        *i3 = (int)(g / ((long long)n5 * n4));
    }
};

// The one-shot combination range ("combo_begin" / "combo_end") belongs to THE NEXT sweep call, whatever becomes of it: the
// entry points take it off the handle before any validation (a call that failed early used to leave it armed
// for an unrelated later sweep) and hand it to check_lags.
struct ComboRange {
    long long begin = 0, end = 0;
};
ComboRange take_combo_range(coreg_handle* h) {
    ComboRange r;
    r.begin = h->opt_combo_begin;
    r.end = h->opt_combo_end;
    h->opt_combo_begin = h->opt_combo_end = 0;
    return r;
}
int check_lags(coreg_handle* h, const coreg_lags* l, LagDims* d, int64_t begin, int64_t end, ComboRange combo = ComboRange()) {
    if (!l || !l->crval1 || !l->crval2 || !l->cdelt1 || !l->cdelt2 || !l->crota)
        return fail(h, COREG_EINVAL, "lags: null array");
    if (l->n_crval1 < 1 || l->n_crval2 < 1 || l->n_cdelt1 < 1 || l->n_cdelt2 < 1 || l->n_crota < 1)
        return fail(h, COREG_EINVAL, "lags: every axis needs at least one value");
    d->n1 = l->n_crval1;
    d->n2 = l->n_crval2;
    d->n3 = l->n_cdelt1;
    d->n4 = l->n_cdelt2;
    d->n5 = l->n_crota;
    d->nc = (long long)d->n3 * d->n4 * d->n5;
    d->c0 = 0;
    {
        // one-shot combination range of a multi-GPU sweep (taken off the handle by the entry point)
        const long long cb = combo.begin, ce = combo.end;
        if (cb != 0 || ce != 0) {
            if (cb < 0 || ce <= cb || ce > d->nc)
                return fail(h, COREG_EINVAL, "combo_begin/combo_end outside [0, n_cdelt1 * n_cdelt2 * n_crota]");
            d->c0 = cb;
            d->nc = ce - cb;
        }
    }
    if (begin < 0 || end > d->total() || begin > end)
        return fail(h, COREG_EINVAL, "lag_begin/lag_end outside [0, n_lags]");
    const double* ax[5] = {l->crval1, l->crval2, l->cdelt1, l->cdelt2, l->crota};
    const int32_t na[5] = {l->n_crval1, l->n_crval2, l->n_cdelt1, l->n_cdelt2, l->n_crota};
    for (int k = 0; k < 5; ++k)
        for (int32_t i = 0; i < na[k]; ++i)
            if (!std::isfinite(ax[k][i])) return fail(h, COREG_EINVAL, "lags: a non-finite value");
    return COREG_OK;
}

// Headers / grids that cannot give finite pixel coordinates are refused before anything is planned or launched
// (geometry.hpp wcs_problem: the reference would hand such a header to astropy, which raises, or return NaN everywhere).
static int check_wcs(coreg_handle* h, const coreg_wcs2d* w, bool carrington_transform) {
    const char* why = wcs_problem(*w, carrington_transform);
    return why ? fail(h, COREG_EINVAL, why) : COREG_OK;
}
// Pixel and grid-point counts are 32-bit in the kernels' lists (active points, border pixels, tiles): refuse what does not fit.
static bool too_many(long long a, long long b) { return a * b > 2147483647ll; }
static int check_grid(coreg_handle* h, const coreg_carr_grid* g) {
    if (g->n_lon >= 1 && g->n_lat >= 1 && too_many(g->n_lon, g->n_lat))
        return fail(h, COREG_EINVAL, "Carrington grid: more than 2^31 - 1 points");
    const char* why = grid_problem(*g);
    return why ? fail(h, COREG_EINVAL, why) : COREG_OK;
}

// Local pixel-space geometry of a sweep (host estimates; they steer the plan, never the results):
// small-image pixels per grid step (d?_di, d?_dj) and per CRVAL1 / CRVAL2 lag step (a?, b?).
struct Geometry {
    double dx_di = 1, dx_dj = 0, dy_di = 0, dy_dj = 1;
    double ax = 0, ay = 0, bx = 0, by = 0;
};
struct Plan {
    int tile_w = 32;      // grid tile = tile_w x (kTilePts / tile_w) points
    int sw = 16, sh = 16; // lag patch of a workgroup: sw CRVAL1 lags x sh CRVAL2 lags (sw * sh <= 256)
    double window = 0;    // estimated LDS window (elements)
    double win_w = 0, win_h = 0;  // its estimated width / height in pixels
};

// Tile shape and lag patch chosen together: fewest lag batches (= least padded lane slots) among the combinations
// whose LDS window (tile extent (+) patch extent, in pixels) fits; ties -> smaller window.  m1 x n2 = lag plane.
Plan choose_plan(coreg_handle* h, const Geometry& g, int m1, int n2, long long lds_elems) {
    Plan best, fallback;
    double best_cost = std::numeric_limits<double>::max(), fb_win = std::numeric_limits<double>::max();
    const int sw_hi = h->opt_patch_w > 0 ? std::min<int>((int)h->opt_patch_w, kBlock) : kBlock;
    for (int tw = 4; tw <= 256; tw *= 2) {
        if (h->opt_tile_w > 0 && tw != h->opt_tile_w) continue;
        const int th = kTilePts / tw;
        if (th < 1) continue;
        const double tex = tw * std::fabs(g.dx_di) + th * std::fabs(g.dx_dj);
        const double tey = tw * std::fabs(g.dy_di) + th * std::fabs(g.dy_dj);
        for (int sw = 1; sw <= std::min(m1, sw_hi); ++sw) {
            int sh = std::min(n2, kBlock / sw);
            if (sh < 1) break;
            const int cols = (m1 + sw - 1) / sw, rows = (n2 + sh - 1) / sh;
            sh = (n2 + rows - 1) / rows;                    // smallest sh with the same batch count
            const int sw2 = (m1 + cols - 1) / cols;         // likewise for sw
            const double ex = tex + sw2 * std::fabs(g.ax) + sh * std::fabs(g.bx) + 6.0;
            const double ey = tey + sw2 * std::fabs(g.ay) + sh * std::fabs(g.by) + 6.0;
            const double win = (ex + 1.0) * ey;
            if (win < fb_win) {
                fb_win = win;
                fallback.tile_w = tw;
                fallback.sw = sw2;
                fallback.sh = sh;
                fallback.window = win;
                fallback.win_w = ex + 1.0;
                fallback.win_h = ey;
            }
            if (win > 0.94 * (double)lds_elems) continue;
            const double cost = (double)cols * rows * (1.0 + 0.08 * win / (double)lds_elems);
            if (cost < best_cost) {
                best_cost = cost;
                best.tile_w = tw;
                best.sw = sw2;
                best.sh = sh;
                best.window = win;
                best.win_w = ex + 1.0;
                best.win_h = ey;
            }
        }
    }
    const Plan& r = best_cost < std::numeric_limits<double>::max() ? best : fallback;
    if (std::getenv("COREG_DEBUG_PLAN"))
        std::fprintf(stderr, "[coreg plan] tile %d x %d, lag patch %d x %d, window estimate %.0f (%.1f x %.1f) of %lld elements; "
                     "geometry d/di (%.3f, %.3f) d/dj (%.3f, %.3f) lag1 (%.3f, %.3f) lag2 (%.3f, %.3f)\n", r.tile_w,
                     kTilePts / r.tile_w, r.sw, r.sh, r.window, r.win_w, r.win_h, lds_elems, g.dx_di, g.dy_di, g.dx_dj, g.dy_dj, g.ax, g.ay,
                     g.bx, g.by);
    return r;
}

// indices of the smallest, the most central and the largest value of a lag axis (any order, NaNs ignored)
void extreme_lags(const double* v, int n, int out[3]) {
    int lo = 0, hi = 0;
    for (int i = 1; i < n; ++i) {
        if (v[i] < v[lo] || v[lo] != v[lo]) lo = i;
        if (v[i] > v[hi] || v[hi] != v[hi]) hi = i;
    }
    const double mid = 0.5 * (v[lo] + v[hi]);
    int m = lo;
    for (int i = 0; i < n; ++i)
        if (std::fabs(v[i] - mid) < std::fabs(v[m] - mid)) m = i;
    out[0] = lo;
    out[1] = m;
    out[2] = hi;
}

// mean spacing of a lag axis over its value range (plan heuristics only; lists may come in any order)
double lag_step(const double* v, int n) {
    if (n < 2) return 0.0;
    double lo = v[0], hi = v[0];
    for (int i = 1; i < n; ++i) {
        lo = std::min(lo, v[i]);
        hi = std::max(hi, v[i]);
    }
    return (hi - lo) / (double)(n - 1);
}

struct SlotList {
    std::vector<int> i1, i2;        // lag indices supplying the lane parameters (clamped for padding)
    std::vector<long long> outidx;  // raveled C-order lag index or -1
    int n_batches = 0;
};

// slots for combo c (= (i3*n4 + i4)*n5 + i5) restricted to the raveled slice [begin, end)
void build_slots(const LagDims& d, long long c, long long begin, long long end, int sw, int sh, SlotList* s) {
    s->i1.clear();
    s->i2.clear();
    s->outidx.clear();
    s->n_batches = 0;
    {
        const size_t cap = (size_t)((d.n1 + sw - 1) / sw + 1) * ((d.n2 + sh - 1) / sh) * kBlock;
        s->i1.reserve(cap);
        s->i2.reserve(cap);
        s->outidx.reserve(cap);
    }
    const long long row = (long long)d.n2 * d.nc;
    const int i1_lo = (int)(begin / row);
    const int i1_hi = (int)((end - 1) / row);
    const int m1 = i1_hi - i1_lo + 1;
    for (int p1 = 0; p1 * sw < m1; ++p1)
        for (int p2 = 0; p2 * sh < d.n2; ++p2) {
            bool any = false;
            const size_t at = s->i1.size();
            for (int t = 0; t < kBlock; ++t) {
                int lx = t % sw, ly = t / sw;
                bool valid = ly < sh;
                if (!valid) lx = ly = 0;
                int i1 = i1_lo + p1 * sw + lx, i2 = p2 * sh + ly;
                if (i1 > i1_hi) {
                    i1 = i1_hi;
                    valid = false;
                }
                if (i2 > d.n2 - 1) {
                    i2 = d.n2 - 1;
                    valid = false;
                }
                const long long idx = ((long long)i1 * d.n2 + i2) * d.nc + c;
                if (idx < begin || idx >= end) valid = false;
                s->i1.push_back(i1);
                s->i2.push_back(i2);
                s->outidx.push_back(valid ? idx : -1);
                any |= valid;
            }
            if (!any) {
                s->i1.resize(at);
                s->i2.resize(at);
                s->outidx.resize(at);
            } else {
                s->n_batches++;
            }
        }
}

int pick_groups(coreg_handle* h, int n_batches, int n_tiles) {
    // the compacted points are cut in n_groups EQUAL shares (k_tile_list), so n_groups * n_batches workgroups of equal
    // work: 256 groups make every round of 256 CUs full; fewer when there are many lag batches
    if (h->opt_shard_world > 1) {
        // point sharding: every rank takes a multiple of 8 groups (the XCD-aware block mapping of k_sweep), as close
        // to one full round of 256 workgroups as the batch count allows
        const long long per_rank = 8 * std::max<long long>(1, std::llround(256.0 / (8.0 * n_batches)));
        return (int)std::min<long long>(per_rank * h->opt_shard_world, 1000 / (8 * h->opt_shard_world) * 8 * h->opt_shard_world);
    }
    long long g = h->opt_n_groups > 0 ? h->opt_n_groups : (4096 + n_batches - 1) / n_batches;
    (void)n_tiles;
    g = std::max<long long>(8, std::min<long long>(h->opt_n_groups > 0 ? 1000 : 256, ((g + 7) / 8) * 8));
    return (int)g;
}

// Tapered shares (kernels.hpp group_start) pay when the launch has many rounds of workgroups; with few rounds the large
// early shares would simply finish last (measured on the translation sweep, profiles/taper_sweep.sh: -3.5 % at 15
// rounds, -4 % at 8, about even at 4, +14 % at 2) ...
void pick_taper(const coreg_handle* h, int n_groups, int n_batches, int* tmin, int* tfrac) {
    *tmin = (int)h->opt_taper_min;
    if (h->opt_taper_frac >= 0) {
        *tfrac = (int)h->opt_taper_frac;
        return;
    }
    // ... and when the groups are many (a fine taper) -- cfg4's 16 groups of a 156-tile grid lost 27 % to it, the
    // 4-round plate-carree launch 20 %
    const double rounds = (double)n_groups * n_batches / 256.0;
    *tfrac = (rounds >= (double)h->opt_taper_rounds && n_groups >= 128 && h->opt_shard_world <= 1) ? 512 : 0;
}

template <int MODE>
int launch_precompute(coreg_handle* h, const PrecomputeArgs& a, int n_tiles, int n_groups, int n_batches) {
    EventPair* ev = next_event(h, h->ev_pre, h->ev_pre_used);
    if (!ev) return fail(h, COREG_EHIP, "hipEventCreate failed");
    int tmin, tfrac;
    pick_taper(h, n_groups, n_batches, &tmin, &tfrac);
    PrecomputeArgs b = a;
    b.prologue = h->pending_prologue;  // (the first launch after upload_plan carries the sweep's prologue)
    const bool with_prologue = b.prologue.src != nullptr;
    std::memset(&h->pending_prologue, 0, sizeof(h->pending_prologue));
    HIPCHK(hipEventRecord(ev->a, h->stream));
    if (h->ref_dtype == COREG_F32)
        hipLaunchKernelGGL((k_precompute<MODE, float>), dim3(n_tiles), dim3(256), 0, h->stream, b);
    else
        hipLaunchKernelGGL((k_precompute<MODE, double>), dim3(n_tiles), dim3(256), 0, h->stream, b);
    (void)with_prologue;
    hipLaunchKernelGGL(k_tile_list, dim3(1), dim3(1024), 0, h->stream, (const int*)a.tile_count, n_tiles, n_groups,
                       h->tile_list.as<int>(), h->tile_cum.as<int>(), h->group_first.as<int>(),
                       h->tile_info.as<long long>(), tmin, tfrac);
    // (no closing event: the sweep launch that follows opens with one, and that is where this interval ends --
    // collect_stats; one marker packet less between the kernels of a sweep)
    ev->b_is_next_sweep = true;
    ev->next_sweep_index = h->ev_sweep_used;
    HIPCHK(hipGetLastError());
    return COREG_OK;
}

int reserve_tiles(coreg_handle* h, int n_tiles) {
    // + an explicit tail: the rolling scalar prefetch of tile_points reads up to kPointGroups chunks past a tile's last
    // chunk (never used), which for the last tile is past the requested size whatever capacity an earlier, larger
    // reservation left
    const size_t pts = (size_t)n_tiles * kTilePts + (size_t)(kPointGroups + 1) * kChunk;
    HIPCHK(h->pts.reserve(pts * sizeof(Pt)));
    HIPCHK(h->tile_count.reserve(n_tiles * sizeof(int)));
    HIPCHK(h->tile_list.reserve(n_tiles * sizeof(int)));
    HIPCHK(h->tile_cum.reserve((n_tiles + 1) * sizeof(int)));
    HIPCHK(h->group_first.reserve(2 * 1024 * sizeof(int) + 64));  // [0, 1024): first list entry; [1024, ...): first unit
    HIPCHK(h->tile_info.reserve(8 * sizeof(long long)));
    HIPCHK(h->tile_bbox.reserve((size_t)n_tiles * 4 * sizeof(double)));
    return COREG_OK;
}

void fill_precompute_common(coreg_handle* h, PrecomputeArgs* a, int tile_w) {
    a->ref = h->ref.p;
    a->gw = h->gW;
    a->gh = h->gH;
    a->tile_w = tile_w;
    a->tile_h = kTilePts / tile_w;
    a->tiles_x = (h->gW + a->tile_w - 1) / a->tile_w;
    a->tiles_y = (h->gH + a->tile_h - 1) / a->tile_h;
    a->pivot_a = h->pivots.as<double>();
    a->pts = h->pts.as<Pt>();
    a->tile_count = h->tile_count.as<int>();
    a->tile_bbox = h->tile_bbox.as<double>();
}

// one sweep-kernel launch + finalize over n_batches * 256 slots whose parameters (SoA [np][n_slots]) and output
// indices are already on the device
long long lds_window_elems(const coreg_handle* h);

// Compile-time LDS window pitch for the Carrington order-2 sweep: the smallest instantiated pitch that holds the planned
// window (width + slack) within the LDS.  All of them are 25 mod 32, the residue that spreads the ~2 px lag lattice best
// over the 32 bank pairs in the conflict simulation.  0 = pitch chosen per visit.
int pick_pitch(const coreg_handle* h, const Plan& plan, long long lds_elems, int order = 2) {
    if (h->opt_pitch == 0 || !h->opt_use_lds) return 0;
    static const int kPitches[] = {89, 121, 153, 185, 217};
    if (h->opt_pitch > 0) {
        for (int p : kPitches)
            if (p == h->opt_pitch) return p;
        return 0;
    }
    for (int p : kPitches)
        if ((double)p >= plan.win_w + (order > 2 ? 4.0 : 2.0) &&
            (double)p * (plan.win_h + (order > 2 ? 2.0 : 0.0)) <= 0.985 * (double)lds_elems)
            return p;
    return 0;
}

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

// Zero-CRVAL lags of a helioprojective sweep share the target header's tangent point, so the map target pixel ->
// shifted pixel is exactly affine, A = (CDELT' PC')^-1 (CDELT PC) about CRPIX.  When A leaves an image axis invariant
// (the zero lag: A = I; a CDELT1-only lag: rows map to rows; CDELT2-only: columns to columns; `reference` CDELT
// semantics: A = I up to the rebuilt PC's last bit) the border rows / columns of the grid sit ON the bounds rule
// c < 0 or c > n-1 and what the reference does with them is decided by the rounding noise of its wcslib round trip
// (alignment.py:1038-1069).  For such a lag the device map is SNAPPED to the exact invariant form (every border
// pixel in bounds along that axis) and the pixels wcslib drops are listed for k_border_fix.
struct AxisInvariance {
    bool rows = false, cols = false;
};
AxisInvariance snap_invariant_axes(const coreg_wcs2d& target, const coreg_wcs2d& shifted, int gw, int gh, double hm[9]) {
    AxisInvariance inv;
    if (target.crval1 != shifted.crval1 || target.crval2 != shifted.crval2 || target.lonpole != shifted.lonpole ||
        target.unit_to_deg != shifted.unit_to_deg)
        return inv;
    const Mat3 a = mat_mul(iwc_to_pix(shifted), pix_to_iwc(target));
    const double tol = 1e-6;  // pixels, over the whole grid; rounding noise is < 1e-9, a real lag moves >> 1e-6
    const double a00 = (double)a.m[0][0], a01 = (double)a.m[0][1], a02 = (double)a.m[0][2];
    const double a10 = (double)a.m[1][0], a11 = (double)a.m[1][1], a12 = (double)a.m[1][2];
    inv.rows = std::fabs(a10) * gw + std::fabs(a11 - 1.0) * gh + std::fabs(a12) < tol;
    inv.cols = std::fabs(a00 - 1.0) * gw + std::fabs(a01) * gh + std::fabs(a02) < tol;
    if (inv.rows || inv.cols) {
        hm[0] = inv.cols ? 1.0 : a00;
        hm[1] = inv.cols ? 0.0 : a01;
        hm[2] = inv.cols ? 0.0 : a02;
        hm[3] = inv.rows ? 0.0 : a10;
        hm[4] = inv.rows ? 1.0 : a11;
        hm[5] = inv.rows ? 0.0 : a12;
        hm[6] = hm[7] = 0.0;
        hm[8] = 1.0;
    }
    return inv;
}

// Pixels of the invariant border rows / columns that the reference's round trip pixel -> sky (target header) ->
// ang2pipi -> pixel (shifted header) sends outside [0, W-1] x [0, H-1] of the image to align.  Appended to `out`.
template <typename Chain>
void wcslib_dropped_border_pixels_t(coreg_handle* h, const coreg_wcs2d& target, const coreg_wcs2d& shifted,
                                    AxisInvariance inv, std::vector<int>* out) {
    const int gw = h->gW, gh = h->gH;
    std::vector<double> key = {(double)target.proj, target.latpole == target.latpole ? target.latpole : -999.0, target.crpix1, target.crpix2, target.crval1, target.crval2, target.cdelt1, target.cdelt2,
                               target.pc1_1, target.pc1_2, target.pc2_1, target.pc2_2, target.unit_to_deg,
                               target.lonpole == target.lonpole ? target.lonpole : -999.0,
                               shifted.crpix1, shifted.crpix2, shifted.cdelt1, shifted.cdelt2, shifted.pc1_1,
                               shifted.pc1_2, shifted.pc2_1, shifted.pc2_2, (double)gw, (double)gh, (double)h->sW,
                               (double)h->sH, inv.rows ? 1.0 : 0.0, inv.cols ? 1.0 : 0.0};
    auto hit = h->border_cache.find(key);
    if (hit == h->border_cache.end()) {
        Chain wf, wt;
        wf.init(target);
        wt.init(shifted);
        std::vector<int> cand;  // row-major, each pixel once
        const int jb = h->sH - 1, ib = h->sW - 1;
        for (int j = 0; j < gh; ++j) {
            const bool row = inv.rows && (j == 0 || j == jb);
            if (row) {
                for (int i = 0; i < gw; ++i) cand.push_back(j * gw + i);
            } else if (inv.cols) {
                cand.push_back(j * gw);
                if (ib > 0 && ib < gw) cand.push_back(j * gw + ib);
            }
        }
        std::vector<char> drop(cand.size(), 0);
        const double wmax = (double)(h->sW - 1), hmax = (double)(h->sH - 1);
        auto work = [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                double x, y;
                wcslib_pixel_to_pixel(wf, wt, (double)(cand[k] % gw), (double)(cand[k] / gw), &x, &y);
                drop[k] = !((x >= 0.0) && (x <= wmax) && (y >= 0.0) && (y <= hmax));  // NaN -> dropped
            }
        };
        unsigned nt = std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
        if (cand.size() < 2048) nt = 1;
        if (nt <= 1) {
            work(0, cand.size());
        } else {
            std::vector<std::thread> th;
            const size_t step = (cand.size() + nt - 1) / nt;
            for (unsigned t = 0; t < nt; ++t) {
                const size_t lo = std::min(cand.size(), (size_t)t * step), hi = std::min(cand.size(), lo + step);
                if (hi > lo) th.emplace_back(work, lo, hi);
            }
            for (auto& x : th) x.join();
        }
        std::vector<int> dropped;
        for (size_t k = 0; k < cand.size(); ++k)
            if (drop[k]) dropped.push_back(cand[k]);
        if (h->border_cache.size() >= 64) h->border_cache.clear();
        hit = h->border_cache.emplace(std::move(key), std::move(dropped)).first;
    }
    out->insert(out->end(), hit->second.begin(), hit->second.end());
}

void wcslib_dropped_border_pixels(coreg_handle* h, const coreg_wcs2d& target, const coreg_wcs2d& shifted,
                                  AxisInvariance inv, std::vector<int>* out) {
    if (target.proj == COREG_PROJ_CAR) wcslib_dropped_border_pixels_t<WcslibCar>(h, target, shifted, inv, out);
    else wcslib_dropped_border_pixels_t<WcslibTan>(h, target, shifted, inv, out);
}

// Odd spline orders: for every grid pixel, does the reference's round trip come back BELOW the integer along an
// invariant axis (bit 0: rows / y, bit 1: columns / x)?  Then floor(c) -- the first tap of an odd-order spline -- is one
// less than at the exact integer the sweep used (k_parity_fix).  W x H evaluations of the wcslib chain, in threads;
// cached per header pair.
template <typename Chain>
const std::vector<unsigned char>& wcslib_tap_shift_flags_t(coreg_handle* h, const coreg_wcs2d& target,
                                                           const coreg_wcs2d& shifted, AxisInvariance inv) {
    const int gw = h->gW, gh = h->gH;
    std::vector<double> key = {(double)target.proj, target.latpole == target.latpole ? target.latpole : -999.0, target.crpix1, target.crpix2, target.crval1, target.crval2, target.cdelt1, target.cdelt2,
                               target.pc1_1, target.pc1_2, target.pc2_1, target.pc2_2, target.unit_to_deg,
                               target.lonpole == target.lonpole ? target.lonpole : -999.0,
                               shifted.crpix1, shifted.crpix2, shifted.cdelt1, shifted.cdelt2, shifted.pc1_1,
                               shifted.pc1_2, shifted.pc2_1, shifted.pc2_2, (double)gw, (double)gh, (double)h->sW,
                               (double)h->sH, inv.rows ? 1.0 : 0.0, inv.cols ? 1.0 : 0.0};
    auto hit = h->flags_cache.find(key);
    if (hit != h->flags_cache.end()) return hit->second;
    Chain wf, wt;
    wf.init(target);
    wt.init(shifted);
    std::vector<unsigned char> flags((size_t)gw * gh, 0);
    const double wmax = (double)(h->sW - 1), hmax = (double)(h->sH - 1);
    auto work = [&](int j0, int j1) {
        for (int j = j0; j < j1; ++j)
            for (int i = 0; i < gw; ++i) {
                double x, y;
                wcslib_pixel_to_pixel(wf, wt, (double)i, (double)j, &x, &y);
                unsigned char f = 0;
                if (inv.rows && y < (double)j) f |= 1;
                if (inv.cols && x < (double)i) f |= 2;
                // the bounds rule drops the pixel altogether (border pixels only; k_border_fix has taken it out)
                if (!((x >= 0.0) && (x <= wmax) && (y >= 0.0) && (y <= hmax))) f |= 4;
                flags[(size_t)j * gw + i] = f;
            }
    };
    unsigned nt = std::min<unsigned>(12, std::max(1u, std::thread::hardware_concurrency()));
    if ((long long)gw * gh < 4096) nt = 1;
    if (nt <= 1) {
        work(0, gh);
    } else {
        std::vector<std::thread> th;
        const int step = (gh + (int)nt - 1) / (int)nt;
        for (unsigned t = 0; t < nt; ++t) {
            const int lo = std::min(gh, (int)t * step), hi = std::min(gh, lo + step);
            if (hi > lo) th.emplace_back(work, lo, hi);
        }
        for (auto& x : th) x.join();
    }
    if (h->flags_cache.size() >= 4) h->flags_cache.clear();
    return h->flags_cache.emplace(std::move(key), std::move(flags)).first->second;
}

const std::vector<unsigned char>& wcslib_tap_shift_flags(coreg_handle* h, const coreg_wcs2d& target,
                                                         const coreg_wcs2d& shifted, AxisInvariance inv) {
    if (target.proj == COREG_PROJ_CAR) return wcslib_tap_shift_flags_t<WcslibCar>(h, target, shifted, inv);
    return wcslib_tap_shift_flags_t<WcslibTan>(h, target, shifted, inv);
}

int upload_border_pixels(coreg_handle* h, const std::vector<int>& pixels) {
    const size_t bytes = std::max<size_t>(1, pixels.size()) * sizeof(int);
    HIPCHK(h->border_dev.reserve(bytes));
    HIPCHK(hipStreamSynchronize(h->stream));  // an earlier sweep may still read the old list / the staging buffer
    HIPCHK(h->pin_border.reserve(bytes));
    std::memcpy(h->pin_border.p, pixels.data(), pixels.size() * sizeof(int));
    if (!pixels.empty())
        HIPCHK(hipMemcpyAsync(h->border_dev.p, h->pin_border.p, pixels.size() * sizeof(int), hipMemcpyHostToDevice,
                              h->stream));
    return COREG_OK;
}

// Odd spline orders, general case (kernels.hpp k_tap_scan / k_tap_fix).  Called between the precompute launch (whose
// prologue has put the lag parameters on the device) and the sweep launch: list the (slot, pixel) samples whose mapped
// coordinate lies within 1e-8 px of an integer, evaluate wcslib's chain for them on the host (`shifted_of(slot)` gives the
// slot's shifted header), and leave everything k_tap_fix needs on the device.  The list is sorted (slot, pixel): one
// workgroup per slot adds its entries in a fixed order.  A list beyond "tap_cap" entries (a pure CRVAL1 / CRVAL2 lag set
// under an unrotated header at full size) is not applied at all -- recorded in tap_last, coreg_last_tap_fix.
template <typename ShiftedOf>
// `hom_dev`: the launch's lane parameters (null: the handle's whole buffer); `car_inv` / `car_fwd`: MODE_CAR launches.
int prepare_tap_fix(coreg_handle* h, int sweep_mode, int order, const coreg_wcs2d& target, long long n_slots,
                    const std::vector<unsigned char>& skip, const double box[4], ShiftedOf shifted_of, BorderFix* fix,
                    const double* hom_dev = nullptr, const LaunchU* car_inv = nullptr, const LaunchU* car_fwd = nullptr) {
    // the list starts small (64 K entries, or what an earlier sweep needed) and is grown -- and the scan repeated --
    // only when a sweep lists more, up to "tap_cap"
    const unsigned cap_max = (unsigned)h->opt_tap_cap;
    unsigned cap = (unsigned)std::min<size_t>(cap_max, std::max<size_t>((size_t)1 << 16, h->tap_list.cap / sizeof(uint2)));
    HIPCHK(h->tap_count.reserve(2 * sizeof(unsigned)));  // [0] listed samples, [1] queued segments
    const unsigned seg_cap = 1u << 20;                   // 16 MiB of (slot, row, first, end); beyond: tested in-thread
    HIPCHK(h->tap_segq.reserve((size_t)seg_cap * sizeof(uint4)));
    HIPCHK(h->tap_list.reserve((size_t)cap * sizeof(uint2)));
    HIPCHK(h->tap_skip.reserve((size_t)n_slots));
    HIPCHK(hipMemcpyAsync(h->tap_skip.p, skip.data(), (size_t)n_slots, hipMemcpyHostToDevice, h->stream));
    TapScanArgs a;
    std::memset(&a.cu, 0, sizeof(a.cu));
    std::memset(&a.fwd, 0, sizeof(a.fwd));
    if (car_inv) a.cu = *car_inv;
    if (car_fwd) a.fwd = *car_fwd;
    a.hom = hom_dev ? hom_dev : h->lane_params.as<double>();
    a.n_slots = n_slots;
    a.skip = h->tap_skip.as<unsigned char>();
    a.ref = h->ref.p;
    a.ref_f32 = h->ref_dtype == COREG_F32 ? 1 : 0;
    a.gw = h->gW;
    a.gh = h->gH;
    a.wmax = (double)(h->sW - 1);
    a.hmax = (double)(h->sH - 1);
    a.tol = 1e-8;  // wcslib's round-trip noise stays below 1e-9 px, the homography's below 1e-11
    a.img = h->small.p;
    a.img_f32 = h->small_f32 ? 1 : 0;
    a.W = h->sW;
    a.H = h->sH;
    a.order = order;
    a.nan_filter = (int)h->opt_tap_nan_filter;
    a.bounds_only = (order & 1) ? 0 : 1;
    if (a.bounds_only) a.nan_filter = 0;  // (nothing read from the image to align: no join with its upload either)
    a.seg_list = h->tap_segq.as<uint4>();
    a.seg_count = h->tap_count.as<unsigned>() + 1;
    a.seg_cap = seg_cap;
    if (a.nan_filter) RETCHK(join_small(h));  // (the scan reads the image to align)
    a.count = h->tap_count.as<unsigned>();
    a.list = h->tap_list.as<uint2>();
    a.cap = cap;
    // the sweep's cull box (target pixels that can map into the image for some lag, 3 px of margin): [x0, x1, y0, y1]
    a.i_lo = (int)std::max(0.0, std::min((double)h->gW, box[0]));
    a.i_hi = (int)std::min((double)(h->gW - 1), std::max(-1.0, box[1]));
    a.j_lo = (int)std::max(0.0, std::min((double)h->gH, box[2]));
    a.j_hi = (int)std::min((double)(h->gH - 1), std::max(-1.0, box[3]));
    h->tap_last[0] = h->tap_last[1] = h->tap_last[2] = 0;
    if (a.i_hi < a.i_lo || a.j_hi < a.j_lo) return COREG_OK;
    const int n_rows = a.j_hi - a.j_lo + 1;
    const unsigned gx = (unsigned)((n_slots + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(n_rows, (int)(4096 / std::max(1u, gx))));
    a.rows_per_block = (n_rows + (int)gy - 1) / (int)gy;
    const dim3 grid(gx, (unsigned)((n_rows + a.rows_per_block - 1) / a.rows_per_block));
    unsigned count = 0;
    for (int pass = 0; pass < 2; ++pass) {
        a.list = h->tap_list.as<uint2>();
        a.cap = cap;
        HIPCHK(hipMemsetAsync(h->tap_count.p, 0, 2 * sizeof(unsigned), h->stream));
        if (sweep_mode == MODE_CAR) {
            hipLaunchKernelGGL((k_tap_scan<MODE_CAR>), grid, dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((k_tap_scan_segments<MODE_CAR>), dim3(2048), dim3(256), 0, h->stream, a);
        } else if (sweep_mode == MODE_HOMOGRAPHY_SERIES) {
            hipLaunchKernelGGL((k_tap_scan<MODE_HOMOGRAPHY_SERIES>), grid, dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((k_tap_scan_segments<MODE_HOMOGRAPHY_SERIES>), dim3(2048), dim3(256), 0, h->stream, a);
        } else {
            hipLaunchKernelGGL((k_tap_scan<MODE_HOMOGRAPHY>), grid, dim3(256), 0, h->stream, a);
            hipLaunchKernelGGL((k_tap_scan_segments<MODE_HOMOGRAPHY>), dim3(2048), dim3(256), 0, h->stream, a);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&count, h->tap_count.p, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (count <= cap || count > cap_max) break;
        cap = count;  // (the scan is deterministic in what it lists: the second pass finds exactly `count` entries)
        HIPCHK(h->tap_list.reserve((size_t)cap * sizeof(uint2)));
    }
    h->tap_last[0] = count;
    h->tap_last[1] = 0;
    h->tap_last[2] = count > cap ? 1 : 0;
    if (count == 0 || count > cap) return COREG_OK;
    std::vector<uint2> list(count);
    HIPCHK(hipMemcpy(list.data(), h->tap_list.p, (size_t)count * sizeof(uint2), hipMemcpyDeviceToHost));
    // group by slot (counting sort), then every segment is put in pixel order by the thread that evaluates it: the
    // summation order of k_tap_fix does not depend on the order the scan's atomics happened to list the samples in
    std::vector<int> first((size_t)n_slots + 1, 0);
    for (unsigned k = 0; k < count; ++k) ++first[(size_t)list[k].x + 1];
    for (long long sl = 0; sl < n_slots; ++sl) first[(size_t)sl + 1] += first[(size_t)sl];
    std::vector<unsigned> pixel(count);
    {
        std::vector<int> at(first.begin(), first.end() - 1);
        for (unsigned k = 0; k < count; ++k) pixel[(size_t)at[list[k].x]++] = list[k].y;
    }
    std::vector<int> seg_slot, seg_begin;
    for (long long sl = 0; sl < n_slots; ++sl)
        if (first[(size_t)sl + 1] > first[(size_t)sl]) {
            seg_slot.push_back((int)sl);
            seg_begin.push_back(first[(size_t)sl]);
        }
    seg_begin.push_back((int)count);
    const int n_seg = (int)seg_slot.size();
    std::vector<double> xw(count), yw(count);
    WcslibTan wf;
    WcslibCar wfc;
    const bool car = target.proj == COREG_PROJ_CAR;
    if (car) wfc.init(target);
    else wf.init(target);
    const int gw = h->gW;
    auto work = [&](int s0, int s1) {
        for (int sg = s0; sg < s1; ++sg) {
            std::sort(pixel.begin() + seg_begin[sg], pixel.begin() + seg_begin[sg + 1]);
            if (car) {
                WcslibCar wt;
                wt.init(shifted_of(seg_slot[sg]));
                for (int e = seg_begin[sg]; e < seg_begin[sg + 1]; ++e)
                    wcslib_pixel_to_pixel(wfc, wt, (double)(pixel[e] % (unsigned)gw), (double)(pixel[e] / (unsigned)gw), &xw[e], &yw[e]);
            } else {
                WcslibTan wt;
                wt.init(shifted_of(seg_slot[sg]));
                for (int e = seg_begin[sg]; e < seg_begin[sg + 1]; ++e)
                    wcslib_pixel_to_pixel(wf, wt, (double)(pixel[e] % (unsigned)gw), (double)(pixel[e] / (unsigned)gw), &xw[e], &yw[e]);
            }
        }
    };
    unsigned nt = std::min<unsigned>(12, std::max(1u, std::thread::hardware_concurrency()));
    if (count < 4096 || n_seg < 2) nt = 1;
    if (nt <= 1) {
        work(0, n_seg);
    } else {
        // segments dealt in runs of about equal entry counts
        std::vector<std::thread> th;
        int s0 = 0;
        for (unsigned t = 0; t < nt && s0 < n_seg; ++t) {
            const long long want = (long long)count * (t + 1) / nt;
            int s1 = s0 + 1;
            while (s1 < n_seg && seg_begin[s1] < want) ++s1;
            if (t + 1 == nt) s1 = n_seg;
            th.emplace_back(work, s0, s1);
            s0 = s1;
        }
        for (auto& x : th) x.join();
    }
    HIPCHK(h->tap_seg_slot.reserve((size_t)n_seg * sizeof(int)));
    HIPCHK(h->tap_seg_begin.reserve((size_t)(n_seg + 1) * sizeof(int)));
    HIPCHK(h->tap_pixel.reserve((size_t)count * sizeof(unsigned)));
    HIPCHK(h->tap_xw.reserve((size_t)count * sizeof(double)));
    HIPCHK(h->tap_yw.reserve((size_t)count * sizeof(double)));
    // (pageable sources, rare path: blocking copies)
    HIPCHK(hipMemcpy(h->tap_seg_slot.p, seg_slot.data(), (size_t)n_seg * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_seg_begin.p, seg_begin.data(), (size_t)(n_seg + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_pixel.p, pixel.data(), (size_t)count * sizeof(unsigned), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_xw.p, xw.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tap_yw.p, yw.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    fix->tap_segs = n_seg;
    fix->tap_count = (long long)count;
    fix->tap_mode = sweep_mode;
    fix->tap.seg_slot = h->tap_seg_slot.as<int>();
    fix->tap.seg_begin = h->tap_seg_begin.as<int>();
    fix->tap.pixel = h->tap_pixel.as<unsigned>();
    fix->tap.xw = h->tap_xw.as<double>();
    fix->tap.yw = h->tap_yw.as<double>();
    fix->tap.cu = a.cu;
    fix->tap.fwd = a.fwd;
    h->tap_last[1] = n_seg;
    return COREG_OK;
}

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

int coreg_set_small(coreg_handle* h, const double* img, int32_t ny, int32_t nx) {
    if (!h) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || too_many(ny, nx)) return fail(h, COREG_EINVAL, "set_small: bad image");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx;
    RETCHK(upload_image(h, img, n, h->small, &h->small_f32));
    h->sW = nx;
    h->sH = ny;
    if (h->small_f32)
        RETCHK(device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1));
    else
        RETCHK(device_mean<double>(h, h->small.as<double>(), (long long)n, h->pivots.as<double>() + 1));
    return COREG_OK;
}

int coreg_set_small_f32(coreg_handle* h, const float* img, int32_t ny, int32_t nx) {
    if (!h) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || too_many(ny, nx)) return fail(h, COREG_EINVAL, "set_small_f32: bad image");
    trace("set_small_f32: enter");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx;
    HIPCHK(h->small.reserve(n * sizeof(float)));
    // through pinned staging: the caller's buffer is free again on return, the copy itself is asynchronous -- and on the
    // upload stream, so that a reference preparation called next does not wait for it
    hipStream_t s;
    RETCHK(begin_small_upload(h, &s));
    if (h->opt_async_upload && s != h->stream) {
        h->small_f32 = true;
        h->sW = nx;
        h->sH = ny;
        void* dev = h->small.p;
        post_upload(h, [h, dev, img, n, s] { return upload_small_worker(h, dev, img, n, false, s); });
        h->small_pending = true;  // (join_small: waits for the worker to have issued everything, then for ev_small)
        trace("set_small_f32: handed to the upload thread");
        return COREG_OK;
    }
    RETCHK(staged_upload(h, h->small.p, img, n * sizeof(float), s));
    h->small_f32 = true;
    h->sW = nx;
    h->sH = ny;
    RETCHK(device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1, s));
    trace("set_small_f32: issued");
    return end_small_upload(h, s);
}

// image to align from pinned host memory or from this GPU's memory (one asynchronous copy, no staging)
static int set_small_direct(coreg_handle* h, const void* img, int dtype, int32_t ny, int32_t nx, SrcKind kind) {
    if (!h) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || too_many(ny, nx) || (dtype != COREG_F32 && dtype != COREG_F64))
        return fail(h, COREG_EINVAL, "set_small: bad argument");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx;
    if (dtype == COREG_F32) {
        HIPCHK(h->small.reserve(n * sizeof(float)));
        HIPCHK(hipMemcpyAsync(h->small.p, img, n * sizeof(float),
                              kind == SRC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        h->small_f32 = true;
    } else {
        RETCHK(upload_image(h, (const double*)img, n, h->small, &h->small_f32, kind));
    }
    h->sW = nx;
    h->sH = ny;
    if (h->small_f32) return device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1);
    return device_mean<double>(h, h->small.as<double>(), (long long)n, h->pivots.as<double>() + 1);
}

int coreg_set_small_from_device(coreg_handle* h, const void* dev_img, int dtype, int32_t ny, int32_t nx) {
    return set_small_direct(h, dev_img, dtype, ny, nx, SRC_DEVICE);
}

// image to align as the FITS data unit stores it (host or page-locked memory): raw bytes up, decode on the GPU
static int set_small_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx, SrcKind kind) {
    if (!h) return COREG_EINVAL;
    PixFmt fmt;
    RETCHK(check_fits(h, px, &fmt));
    if (ny < 1 || nx < 1 || too_many(ny, nx)) return fail(h, COREG_EINVAL, "set_small_fits: bad image size");
    RETCHK(bind_device(h));
    const size_t n = (size_t)ny * nx, eb = fmt.elem();
    DevBuf& dst = fmt.swap_only() ? h->small : h->up_raw;
    HIPCHK(dst.reserve(n * eb));
    if (fmt.swap_only() && kind == SRC_HOST) {
        // BITPIX = -32 from host memory (what an EUI level-2 file without tile compression holds): upload stream
        hipStream_t s;
        RETCHK(begin_small_upload(h, &s));
        if (h->opt_async_upload && s != h->stream) {
            h->small_f32 = true;
            h->sW = nx;
            h->sH = ny;
            void* dev = dst.p;
            const void* src = px->data;
            post_upload(h, [h, dev, src, n, s] { return upload_small_worker(h, dev, src, n, true, s); });
            h->small_pending = true;
            return COREG_OK;
        }
        RETCHK(staged_upload(h, dst.p, px->data, n * eb, s));
        const int nb = (int)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_fits_swap32, dim3(nb), dim3(256), 0, s, (unsigned int*)dst.p, (long long)n);
        HIPCHK(hipGetLastError());
        h->small_f32 = true;
        h->sW = nx;
        h->sH = ny;
        RETCHK(device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1, s));
        return end_small_upload(h, s);
    }
    if (kind == SRC_PINNED) HIPCHK(hipMemcpyAsync(dst.p, px->data, n * eb, hipMemcpyHostToDevice, h->stream));
    else if (kind == SRC_DEVICE) HIPCHK(hipMemcpyAsync(dst.p, px->data, n * eb, hipMemcpyDeviceToDevice, h->stream));
    else RETCHK(staged_upload(h, dst.p, px->data, n * eb));
    RETCHK(fits_decode(h, fmt, dst.p, n, h->small, &h->small_f32));
    h->sW = nx;
    h->sH = ny;
    if (h->small_f32) return device_mean<float>(h, h->small.as<float>(), (long long)n, h->pivots.as<double>() + 1);
    return device_mean<double>(h, h->small.as<double>(), (long long)n, h->pivots.as<double>() + 1);
}

int coreg_set_small_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx) {
    return set_small_fits(h, px, ny, nx, SRC_HOST);
}

int coreg_set_small_tiled(coreg_handle* h, const coreg_fits_tiled* t) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    RETCHK(decode_tiled_device(h, t, h->small, &h->small_f32));
    h->sW = t->naxis1;
    h->sH = t->naxis2;
    const long long n = (long long)h->sW * h->sH;
    if (h->small_f32) return device_mean<float>(h, h->small.as<float>(), n, h->pivots.as<double>() + 1);
    return device_mean<double>(h, h->small.as<double>(), n, h->pivots.as<double>() + 1);
}

int coreg_decode_tiled_host(const coreg_fits_tiled* t, void* out, int dtype, int32_t* tile_status) {
    if (check_tiled(t) || !out || (dtype != COREG_F32 && dtype != COREG_F64)) return COREG_EINVAL;
    if (dtype == COREG_F32 && t->zbitpix != -32) return COREG_EINVAL;
    static const std::vector<float> randoms = [] {
        std::vector<float> r(coregrice::kNRandom);
        coregrice::init_randoms(r.data());
        return r;
    }();
    coregrice::TileImage im;
    fill_tile_image(*t, &im);
    im.heap = (const unsigned char*)t->heap;
    im.tile_offset = t->tile_offset;
    im.tile_nbytes = t->tile_nbytes;
    im.zscale = t->zscale;
    im.zzero = t->zzero;
    im.randoms = randoms.data();
    im.out = out;
    im.out_dtype = dtype == COREG_F32 ? coregrice::OUT_F32 : coregrice::OUT_F64;
    const int nt = t->n_tiles;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n_thr = (int)std::min<long long>(std::min<unsigned>(hw, 12u),
                                               std::max<long long>(1, (long long)t->naxis1 * t->naxis2 / (1 << 16)));
    auto work = [&](int lo, int hi) {
        for (int k = lo; k < hi; ++k) {
            const int e = coregrice::decode_tile(im, k);
            if (tile_status) tile_status[k] = e;
        }
    };
    if (n_thr <= 1) {
        work(0, nt);
        return COREG_OK;
    }
    std::vector<std::thread> th;
    const int per = (nt + n_thr - 1) / n_thr;
    for (int k = 1; k < n_thr; ++k) th.emplace_back(work, std::min(nt, k * per), std::min(nt, (k + 1) * per));
    work(0, std::min(nt, per));
    for (auto& x : th) x.join();
    return COREG_OK;
}

int coreg_encode_tiled_host(const void* pixels, int dtype, int ny, int nx, int tile_x, int tile_y, int bytepix,
                            int blocksize, int quantize, int dither0, double scale, unsigned char* heap,
                            long long heap_cap, int32_t* tile_nbytes, int64_t* tile_offset, double* zscale, double* zzero,
                            long long* heap_used) {
    if (!pixels || !heap || !tile_nbytes || !tile_offset || !heap_used || ny <= 0 || nx <= 0 || tile_x <= 0 || tile_y <= 0)
        return COREG_EINVAL;
    if (blocksize <= 0 || blocksize > 1024 || (bytepix != 1 && bytepix != 2 && bytepix != 4)) return COREG_EINVAL;
    const bool is_float = dtype == COREG_F32 || dtype == COREG_F64;
    if (is_float) {
        if (quantize < coregrice::Q_NO_DITHER || quantize > coregrice::Q_DITHER_2 || !(scale > 0) || !std::isfinite(scale) ||
            !zscale || !zzero || bytepix != 4)
            return COREG_EINVAL;
    } else if (dtype != COREG_I32) {
        return COREG_EINVAL;  // integer images: the stored integers as int32, whatever BYTEPIX
    }
    static const std::vector<float> randoms = [] {
        std::vector<float> r(coregrice::kNRandom);
        coregrice::init_randoms(r.data());
        return r;
    }();
    coregrice::TileImage t{};
    t.naxis1 = nx;
    t.naxis2 = ny;
    t.ztile1 = tile_x;
    t.ztile2 = tile_y;
    t.dither0 = dither0;
    const int ntx = (nx + tile_x - 1) / tile_x, nty = (ny + tile_y - 1) / tile_y;
    std::vector<int32_t> q((size_t)tile_x * tile_y);
    long long used = 0;
    for (int n = 0; n < ntx * nty; ++n) {
        const coregrice::TileBox b = coregrice::tile_box(t, n);
        const int npx = b.tw * b.th;
        if (is_float) {
            const int iseed = coregrice::dither_seed(t, n);
            const int e = dtype == COREG_F32
                              ? coregrice::quantize_tile((const float*)pixels, nx, b, quantize, iseed, randoms.data(), scale,
                                                         q.data(), &zzero[n])
                              : coregrice::quantize_tile((const double*)pixels, nx, b, quantize, iseed, randoms.data(), scale,
                                                         q.data(), &zzero[n]);
            if (e) return COREG_EINVAL;  // (the tile's range does not fit 32-bit integers at this scale)
            zscale[n] = scale;
        } else {
            const int32_t* src = (const int32_t*)pixels;
            for (int y = 0; y < b.th; ++y)
                std::memcpy(q.data() + (size_t)y * b.tw, src + (size_t)(b.y0 + y) * nx + b.x0, (size_t)b.tw * 4);
        }
        const int64_t len = coregrice::rice_encode_tile(q.data(), npx, blocksize, bytepix, heap + used, heap_cap - used);
        if (len < 0) return COREG_ENOMEM;
        tile_offset[n] = used;
        tile_nbytes[n] = (int32_t)len;
        used += len;
    }
    *heap_used = used;
    return COREG_OK;
}

int coreg_threshold_small(coreg_handle* h, int has_min, double vmin, int has_max, double vmax, long long* n_finite) {
    if (!h) return COREG_EINVAL;
    if (!h->small.p) return fail(h, COREG_ESTATE, "coreg_set_small has not been called");
    RETCHK(bind_device(h));
    const long long n = (long long)h->sW * h->sH;
    if (has_min || has_max) {
        const int nb = (int)std::min<long long>((n + 255) / 256, 2048);
        if (h->small_f32)
            hipLaunchKernelGGL((k_threshold<float>), dim3(nb), dim3(256), 0, h->stream, h->small.as<float>(), n, has_min,
                               vmin, has_max, vmax);
        else
            hipLaunchKernelGGL((k_threshold<double>), dim3(nb), dim3(256), 0, h->stream, h->small.as<double>(), n, has_min,
                               vmin, has_max, vmax);
        HIPCHK(hipGetLastError());
    }
    // pivot = mean of what is left (same value as uploading a host-thresholded image)
    if (h->small_f32)
        RETCHK(device_mean<float>(h, h->small.as<float>(), n, h->pivots.as<double>() + 1));
    else
        RETCHK(device_mean<double>(h, h->small.as<double>(), n, h->pivots.as<double>() + 1));
    if (n_finite) {
        long long cnt[256];
        HIPCHK(hipMemcpyAsync(cnt, h->red_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        long long c = 0;
        for (int i = 0; i < 256; ++i) c += cnt[i];
        *n_finite = c;
    }
    return COREG_OK;
}

static int ref_pivot(coreg_handle* h) {
    const long long n = (long long)h->gW * h->gH;
    if (h->ref_dtype == COREG_F32) return device_mean<float>(h, h->ref.as<float>(), n, h->pivots.as<double>());
    return device_mean<double>(h, h->ref.as<double>(), n, h->pivots.as<double>());
}

int coreg_set_reference_on_grid(coreg_handle* h, const void* ref, int dtype, int32_t gy, int32_t gx) {
    if (!h) return COREG_EINVAL;
    if (!ref || gy < 1 || gx < 1 || (dtype != COREG_F32 && dtype != COREG_F64))
        return fail(h, COREG_EINVAL, "set_reference_on_grid: bad argument");
    RETCHK(bind_device(h));
    const size_t bytes = (size_t)gy * gx * (dtype == COREG_F32 ? 4 : 8);
    HIPCHK(h->ref.reserve(bytes));
    HIPCHK(hipMemcpyAsync(h->ref.p, ref, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->gW = gx;
    h->gH = gy;
    h->ref_dtype = dtype;
    return ref_pivot(h);
}

// Which pixels of the W x H source image can the once-only resample touch?  The bounding box of the in-bounds sample
// coordinates is computed on the GPU by the same coordinate function the resample uses (k_resample_bbox), widened by the
// spline apron and clipped to the image.  Uploading only that rectangle -- the Carrington grid of the headline touches
// 2 % of the 3072 x 3072 reference, the sub-map of a 2048 x 2048 HRIEUV field 0.6 % -- takes the reference image out of
// the PCIe-inclusive cost of a call; results are bit-identical (same pixels, same arithmetic).  Costs one ~20 us kernel
// and a 4-double read-back.  crop = {0, 0, W, H} when cropping would not pay (more than half the image) or is disabled
// (coreg_set_option "crop_reference" 0).
struct CropRect {
    int x0, y0, w, h;
};
static int reference_crop(coreg_handle* h, int mode, const ResampleArgs& a0, int order, CropRect* out) {
    *out = {0, 0, a0.W, a0.H};
    if (!h->opt_crop_reference || a0.W < 64 || a0.H < 64) return COREG_OK;
    const int nb = 256;
    // On a side stream: the box depends on headers and grid tables only, so it need not queue behind the upload of the
    // image to align that usually precedes it on the handle's stream (its last DMA segment is still in flight).
    if (!h->aux_stream) HIPCHK(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    HIPCHK(h->bbox_buf.reserve((size_t)nb * 4 * sizeof(double)));
    ResampleArgs a = a0;
    a.bbox = h->bbox_buf.as<double>();
    if (mode == MODE_TRANSLATE)
        hipLaunchKernelGGL((k_resample_bbox<MODE_TRANSLATE>), dim3(nb), dim3(256), 0, h->aux_stream, a);
    else if (mode == MODE_CAR)
        hipLaunchKernelGGL((k_resample_bbox<MODE_CAR>), dim3(nb), dim3(256), 0, h->aux_stream, a);
    else
        hipLaunchKernelGGL((k_resample_bbox<MODE_HOMOGRAPHY>), dim3(nb), dim3(256), 0, h->aux_stream, a);
    HIPCHK(hipGetLastError());
    std::vector<double> part((size_t)nb * 4);
    HIPCHK(hipMemcpyAsync(part.data(), a.bbox, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->aux_stream));
    HIPCHK(hipStreamSynchronize(h->aux_stream));
    double mnx = 1e300, mxx = -1e300, mny = 1e300, mxy = -1e300;
    for (int b = 0; b < nb; ++b) {
        mnx = std::min(mnx, part[4 * b + 0]);
        mxx = std::max(mxx, part[4 * b + 1]);
        mny = std::min(mny, part[4 * b + 2]);
        mxy = std::max(mxy, part[4 * b + 3]);
    }
    const int apron = order / 2 + 3;  // taps of an in-bounds sample: [floor(c) - order/2 - 1, floor(c) + order - order/2 + 1]
    int x0 = 0, x1 = apron * 2, y0 = 0, y1 = apron * 2;  // nothing in bounds: any small rectangle (never read)
    if (mnx <= mxx && mny <= mxy) {
        x0 = std::max(0, (int)std::floor(mnx) - apron);
        x1 = std::min(a0.W - 1, (int)std::floor(mxx) + apron + 1);
        y0 = std::max(0, (int)std::floor(mny) - apron);
        y1 = std::min(a0.H - 1, (int)std::floor(mxy) + apron + 1);
    }
    // (taps that mirror at an image edge stay inside: the rectangle starts AT that edge and is at least 2 aprons wide)
    x1 = std::min(a0.W - 1, std::max(x1, x0 + 2 * apron));
    y1 = std::min(a0.H - 1, std::max(y1, y0 + 2 * apron));
    const long long area = (long long)(x1 - x0 + 1) * (y1 - y0 + 1);
    if (2 * area > (long long)a0.W * a0.H) return COREG_OK;  // not worth a strided copy
    *out = {x0, y0, x1 - x0 + 1, y1 - y0 + 1};
    return COREG_OK;
}

// rows y0 .. of a host image, columns x0 .., packed into pinned staging and sent to `dev` (contiguous, pitch = crop width)
static int staged_upload_rect(coreg_handle* h, void* dev, const void* host, size_t elem, int W, const CropRect& c) {
    const char* src = (const char*)host + ((size_t)c.y0 * W + c.x0) * elem;
    if (c.w == W) return staged_upload(h, dev, src, (size_t)c.w * c.h * elem);  // whole rows: one contiguous range
    const int k = h->pin_img_next;
    h->pin_img_next ^= 1;
    if (!h->ev_img[k]) HIPCHK(hipEventCreateWithFlags(&h->ev_img[k], hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(h->ev_img[k]));
    const size_t bytes = (size_t)c.w * c.h * elem;
    HIPCHK(h->pin_img[k].reserve(bytes));
    parallel_copy_rows(h->pin_img[k].p, src, (size_t)c.h, (size_t)c.w * elem, (size_t)W * elem);
    HIPCHK(hipMemcpyAsync(dev, h->pin_img[k].p, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_img[k], h->stream));
    return COREG_OK;
}

// the reference image's pixels: float64 from the caller (tested for float32-exactness on the GPU), the float32
// pixels a BITPIX=-32 FITS file holds (half the PCIe bytes; the reference's float64 cast of them is exact), or the raw
// big-endian elements of the FITS data unit (decoded on the GPU)
// (src_on_device: the pixels are read where they are, by the resample kernel on the handle's stream -- no copy)
static int upload_reference_source(coreg_handle* h, const void* large, size_t n, const PixFmt& fmt, bool* f32,
                                   SrcKind kind, const void** img_dev, int W = 0, const CropRect* crop = nullptr) {
    const bool src_f32 = fmt.f32;
    if (kind == SRC_DEVICE) {
        if (fmt.raw()) return fail(h, COREG_ENOTIMPL, "raw FITS pixels must come from host memory");
        *f32 = src_f32;
        *img_dev = large;
        return COREG_OK;
    }
    const bool cropped = crop && kind == SRC_HOST && (size_t)crop->w * crop->h < n;
    const size_t nc = cropped ? (size_t)crop->w * crop->h : n;
    if (fmt.raw()) {
        // only the rectangle the resample can touch crosses PCIe, as stored in the file; decoded on the device
        const size_t eb = fmt.elem();
        DevBuf& dst = fmt.swap_only() ? h->tmp_img : h->up_raw;
        HIPCHK(dst.reserve(nc * eb));
        if (cropped) RETCHK(staged_upload_rect(h, dst.p, large, eb, W, *crop));
        else if (kind == SRC_PINNED) HIPCHK(hipMemcpyAsync(dst.p, large, n * eb, hipMemcpyHostToDevice, h->stream));
        else RETCHK(staged_upload(h, dst.p, large, n * eb));
        RETCHK(fits_decode(h, fmt, dst.p, nc, h->tmp_img, f32));
        *img_dev = h->tmp_img.p;
        return COREG_OK;
    }
    if (cropped) {
        // only the rectangle the resample can touch crosses PCIe
        if (src_f32) {
            HIPCHK(h->tmp_img.reserve(nc * sizeof(float)));
            RETCHK(staged_upload_rect(h, h->tmp_img.p, large, sizeof(float), W, *crop));
            *f32 = true;
        } else {
            HIPCHK(h->up_f64.reserve(nc * sizeof(double)));
            RETCHK(staged_upload_rect(h, h->up_f64.p, large, sizeof(double), W, *crop));
            RETCHK(upload_image(h, h->up_f64.as<double>(), nc, h->tmp_img, f32, SRC_DEVICE));
        }
        *img_dev = h->tmp_img.p;
        return COREG_OK;
    }
    if (!src_f32) {
        RETCHK(upload_image(h, (const double*)large, n, h->tmp_img, f32, kind));
    } else {
        HIPCHK(h->tmp_img.reserve(n * sizeof(float)));
        if (kind == SRC_PINNED)
            HIPCHK(hipMemcpyAsync(h->tmp_img.p, large, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
        else
            RETCHK(staged_upload(h, h->tmp_img.p, large, n * sizeof(float)));
        *f32 = true;
    }
    *img_dev = h->tmp_img.p;
    return COREG_OK;
}

static int prepare_carrington(coreg_handle* h, const void* large, const PixFmt& fmt, int32_t ny, int32_t nx,
                              const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r, int order,
                              SrcKind kind = SRC_HOST) {
    if (!h) return COREG_EINVAL;
    if (!large || !hdr || !grid || ny < 1 || nx < 1 || too_many(ny, nx))
        return fail(h, COREG_EINVAL, "prepare_reference: bad argument");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr, true));
    RETCHK(check_grid(h, grid));
    if (!std::isfinite(solar_r) || !(solar_r > 0.0)) return fail(h, COREG_EINVAL, "solar_r must be positive");
    trace("prepare_carrington: enter");
    RETCHK(bind_device_nowait(h));  // (touches neither the image to align nor its pivot: no join with the upload stream)
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    RETCHK(upload_carr_tables(h, *grid, *hdr, &a.carr));
    set_carr_common(&a.carr, carr_common(*hdr, solar_r));
    carr_origin(*hdr, &a.x0, &a.y0);
    a.W = nx;
    a.H = ny;
    a.gw = grid->n_lon;
    a.gh = grid->n_lat;
    CropRect crop = {0, 0, nx, ny};
    if (kind == SRC_HOST) RETCHK(reference_crop(h, MODE_TRANSLATE, a, order, &crop));
    trace("prepare_carrington: crop box known");
    bool f32;
    const void* img_dev = nullptr;
    RETCHK(upload_reference_source(h, large, (size_t)ny * nx, fmt, &f32, kind, &img_dev, nx, &crop));
    a.img = img_dev;
    if (crop.w != nx || crop.h != ny) a.crop = {crop.x0, crop.y0, crop.w};
    HIPCHK(h->ref.reserve((size_t)a.gw * a.gh * sizeof(double)));
    a.out = h->ref.p;
    RETCHK(dispatch_resample(h, MODE_TRANSLATE, order, f32, false, a, 0));
    h->gW = a.gw;
    h->gH = a.gh;
    h->ref_dtype = COREG_F64;
    RETCHK(ref_pivot(h));
    trace("prepare_carrington: issued");
    // no host sync: the pinned staging is guarded by staged_upload's own wait, everything else is stream-ordered
    return COREG_OK;  // tmp_img stays allocated: the next preparation re-uses it (hipFree would stall the device)
}

int coreg_prepare_reference_carrington(coreg_handle* h, const double* large, int32_t ny, int32_t nx,
                                       const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r, int order) {
    return prepare_carrington(h, large, PixFmt::native(false), ny, nx, hdr, grid, solar_r, order);
}

int coreg_prepare_reference_carrington_f32(coreg_handle* h, const float* large, int32_t ny, int32_t nx,
                                           const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r,
                                           int order) {
    return prepare_carrington(h, large, PixFmt::native(true), ny, nx, hdr, grid, solar_r, order);
}

static int prepare_helioprojective(coreg_handle* h, const void* large, const PixFmt& fmt, int32_t ny, int32_t nx,
                                   const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order,
                                   SrcKind kind = SRC_HOST) {
    if (!h) return COREG_EINVAL;
    if (!large || !hdr_large || !hdr_small || ny < 1 || nx < 1 || too_many(ny, nx))
        return fail(h, COREG_EINVAL, "prepare_reference: bad argument");
    if (hdr_small->naxis1 < 1 || hdr_small->naxis2 < 1 || too_many(hdr_small->naxis1, hdr_small->naxis2))
        return fail(h, COREG_EINVAL, "hdr_small: NAXIS1/2 missing (or more than 2^31 - 1 pixels)");
    if (hdr_large->proj != hdr_small->proj || (hdr_small->proj != COREG_PROJ_TAN && hdr_small->proj != COREG_PROJ_CAR))
        return fail(h, COREG_ENOTIMPL, "prepare_reference_helioprojective: both headers TAN, or both CAR");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_large, false));
    RETCHK(check_wcs(h, hdr_small, false));
    RETCHK(bind_device_nowait(h));
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    int mode = MODE_HOMOGRAPHY;
    if (hdr_small->proj == COREG_PROJ_CAR) {
        // two Carrington maps (align_using_initial_carrington: both branches of alignment.py:649-651 / :765-767 build the
        // sub-map for this frame too): pixel of hdr_small -> native angles -> sphere rotation -> native angles of hdr_large
        // -> its pixel, the per-lag map of sweep_car with the roles of the two maps exchanged
        mode = MODE_CAR;
        Mat3 r_small, r_large;
        if (car_native_to_celestial(*hdr_small, &r_small) || car_native_to_celestial(*hdr_large, &r_large))
            return fail(h, COREG_EINVAL, "prepare_reference: no valid native pole for this CRVAL2 / LONPOLE (CAR)");
        const Mat3 m = mat_mul(mat_T(r_large), r_small);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) a.hom.h[3 * i + j] = (double)m.m[i][j];
        const Affine2 fwd = car_pix_to_native(*hdr_small), inv = car_native_to_pix(*hdr_large);
        a.car_fwd.m00 = fwd.m00;
        a.car_fwd.m01 = fwd.m01;
        a.car_fwd.m10 = fwd.m10;
        a.car_fwd.m11 = fwd.m11;
        a.car_fwd.b0 = fwd.b0;
        a.car_fwd.b1 = fwd.b1;
        a.car_inv.m00 = inv.m00;
        a.car_inv.m01 = inv.m01;
        a.car_inv.m10 = inv.m10;
        a.car_inv.m11 = inv.m11;
        a.car_inv.b0 = inv.b0;
        a.car_inv.b1 = inv.b1;
    } else {
        homography(*hdr_small, *hdr_large, a.hom.h);  // alignment.py:993: pixels of hdr_cut -> pixels of hdr_large
    }
    a.W = nx;
    a.H = ny;
    a.gw = hdr_small->naxis1;
    a.gh = hdr_small->naxis2;
    CropRect crop = {0, 0, nx, ny};
    if (kind == SRC_HOST) RETCHK(reference_crop(h, mode, a, order, &crop));
    bool f32;
    const void* img_dev = nullptr;
    RETCHK(upload_reference_source(h, large, (size_t)ny * nx, fmt, &f32, kind, &img_dev, nx, &crop));
    a.img = img_dev;
    if (crop.w != nx || crop.h != ny) a.crop = {crop.x0, crop.y0, crop.w};
    HIPCHK(h->ref.reserve((size_t)a.gw * a.gh * sizeof(float)));
    a.out = h->ref.p;
    RETCHK(dispatch_resample(h, mode, order, f32, true, a, 0));
    h->gW = a.gw;
    h->gH = a.gh;
    h->ref_dtype = COREG_F32;
    RETCHK(ref_pivot(h));
    return COREG_OK;
}

int coreg_prepare_reference_helioprojective(coreg_handle* h, const double* large, int32_t ny, int32_t nx,
                                            const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order) {
    return prepare_helioprojective(h, large, PixFmt::native(false), ny, nx, hdr_large, hdr_small, order);
}

int coreg_prepare_reference_helioprojective_f32(coreg_handle* h, const float* large, int32_t ny, int32_t nx,
                                                const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small,
                                                int order) {
    return prepare_helioprojective(h, large, PixFmt::native(true), ny, nx, hdr_large, hdr_small, order);
}

int coreg_prepare_reference_carrington_from_device(coreg_handle* h, const void* dev_large, int dtype, int32_t ny,
                                                   int32_t nx, const coreg_wcs2d* hdr_large,
                                                   const coreg_carr_grid* grid, double solar_r, int order) {
    if (h && dtype != COREG_F32 && dtype != COREG_F64) return fail(h, COREG_EINVAL, "prepare_reference: bad dtype");
    return prepare_carrington(h, dev_large, PixFmt::native(dtype == COREG_F32), ny, nx, hdr_large, grid, solar_r, order,
                              SRC_DEVICE);
}

int coreg_prepare_reference_helioprojective_from_device(coreg_handle* h, const void* dev_large, int dtype, int32_t ny,
                                                        int32_t nx, const coreg_wcs2d* hdr_large,
                                                        const coreg_wcs2d* hdr_small, int order) {
    if (h && dtype != COREG_F32 && dtype != COREG_F64) return fail(h, COREG_EINVAL, "prepare_reference: bad dtype");
    return prepare_helioprojective(h, dev_large, PixFmt::native(dtype == COREG_F32), ny, nx, hdr_large, hdr_small, order,
                                   SRC_DEVICE);
}

int coreg_prepare_reference_carrington_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                            const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid, double solar_r,
                                            int order) {
    if (!h) return COREG_EINVAL;
    PixFmt fmt;
    RETCHK(check_fits(h, px, &fmt));
    return prepare_carrington(h, px->data, fmt, ny, nx, hdr_large, grid, solar_r, order);
}

int coreg_prepare_reference_helioprojective_fits(coreg_handle* h, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                                 const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order) {
    if (!h) return COREG_EINVAL;
    PixFmt fmt;
    RETCHK(check_fits(h, px, &fmt));
    return prepare_helioprojective(h, px->data, fmt, ny, nx, hdr_large, hdr_small, order);
}

int coreg_prepare_reference_carrington_tiled(coreg_handle* h, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                             const coreg_carr_grid* grid, double solar_r, int order) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    bool f32;
    RETCHK(decode_tiled_device(h, t, h->dec_img, &f32));  // the compressed bytes cross PCIe, the pixels never do
    return prepare_carrington(h, h->dec_img.p, PixFmt::native(f32), t->naxis2, t->naxis1, hdr_large, grid, solar_r, order,
                              SRC_DEVICE);
}

int coreg_prepare_reference_helioprojective_tiled(coreg_handle* h, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                                  const coreg_wcs2d* hdr_small, int order) {
    if (!h) return COREG_EINVAL;
    RETCHK(bind_device(h));
    bool f32;
    RETCHK(decode_tiled_device(h, t, h->dec_img, &f32));
    return prepare_helioprojective(h, h->dec_img.p, PixFmt::native(f32), t->naxis2, t->naxis1, hdr_large, hdr_small, order,
                                   SRC_DEVICE);
}

int coreg_get_reference_on_grid(coreg_handle* h, void* out, int dtype) {
    if (!h) return COREG_EINVAL;
    if (!h->ref.p) return fail(h, COREG_ESTATE, "no reference image on the target grid");
    if (!out || dtype != h->ref_dtype) return fail(h, COREG_EINVAL, "get_reference_on_grid: dtype mismatch");
    RETCHK(bind_device(h));
    const size_t bytes = (size_t)h->gW * h->gH * (dtype == COREG_F32 ? 4 : 8);
    HIPCHK(hipMemcpyAsync(out, h->ref.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_resample_carrington(coreg_handle* h, const coreg_wcs2d* hdr, const coreg_carr_grid* grid, double solar_r,
                              int order, double* out) {
    if (!h) return COREG_EINVAL;
    if (!hdr || !grid || !out) return fail(h, COREG_EINVAL, "resample_carrington: bad argument");
    if (!h->small.p) return fail(h, COREG_ESTATE, "coreg_set_small has not been called");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr, true));
    RETCHK(check_grid(h, grid));
    if (!std::isfinite(solar_r) || !(solar_r > 0.0)) return fail(h, COREG_EINVAL, "solar_r must be positive");
    RETCHK(bind_device(h));
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    RETCHK(upload_carr_tables(h, *grid, *hdr, &a.carr));
    set_carr_common(&a.carr, carr_common(*hdr, solar_r));
    carr_origin(*hdr, &a.x0, &a.y0);
    a.img = h->small.p;
    a.W = h->sW;
    a.H = h->sH;
    a.gw = grid->n_lon;
    a.gh = grid->n_lat;
    const size_t bytes = (size_t)a.gw * a.gh * sizeof(double);
    HIPCHK(h->out_dev.reserve(bytes));
    a.out = h->out_dev.p;
    RETCHK(dispatch_resample(h, MODE_TRANSLATE, order, h->small_f32, false, a, 0));
    HIPCHK(hipMemcpyAsync(out, h->out_dev.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

static int resample_helio(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr, int order, void* out,
                          bool out_f32) {
    if (!h) return COREG_EINVAL;
    if (!hdr_target || !hdr || !out) return fail(h, COREG_EINVAL, "resample_helioprojective: bad argument");
    if (!h->small.p) return fail(h, COREG_ESTATE, "coreg_set_small has not been called");
    if (hdr_target->naxis1 < 1 || hdr_target->naxis2 < 1 || too_many(hdr_target->naxis1, hdr_target->naxis2))
        return fail(h, COREG_EINVAL, "hdr_target: NAXIS missing (or more than 2^31 - 1 pixels)");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_target, false));
    RETCHK(check_wcs(h, hdr, false));
    RETCHK(bind_device(h));
    ResampleArgs a;
    std::memset(&a, 0, sizeof(a));
    if (hdr_target->proj != COREG_PROJ_TAN || hdr->proj != COREG_PROJ_TAN)
        return fail(h, COREG_ENOTIMPL, "resample_helioprojective: TAN headers only");
    homography(*hdr_target, *hdr, a.hom.h);  // alignment.py:1022
    a.img = h->small.p;
    a.W = h->sW;
    a.H = h->sH;
    a.gw = hdr_target->naxis1;
    a.gh = hdr_target->naxis2;
    const size_t bytes = (size_t)a.gw * a.gh * (out_f32 ? sizeof(float) : sizeof(double));
    HIPCHK(h->out_dev.reserve(bytes));
    a.out = h->out_dev.p;
    RETCHK(dispatch_resample(h, MODE_HOMOGRAPHY, order, h->small_f32, out_f32, a, 0));
    HIPCHK(hipMemcpyAsync(out, h->out_dev.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_resample_helioprojective(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr, int order,
                                   float* out) {
    return resample_helio(h, hdr_target, hdr, order, out, true);
}

int coreg_resample_helioprojective_f64(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr,
                                       int order, double* out) {
    return resample_helio(h, hdr_target, hdr, order, out, false);
}

int coreg_sweep_carrington(coreg_handle* h, const coreg_wcs2d* hdr_small, const coreg_carr_grid* grid, double solar_r,
                           const coreg_lags* lags, int order, int method, int cdelt_semantics, int64_t lag_begin,
                           int64_t lag_end, double* corr_out, int out_on_device) {
    if (!h) return COREG_EINVAL;
    const ComboRange combo = take_combo_range(h);
    trace("sweep_carrington: enter");
    if (!hdr_small || !grid) return fail(h, COREG_EINVAL, "sweep_carrington: null header/grid");
    if (method != COREG_METHOD_CORRELATION && method != COREG_METHOD_RESIDUS)
        return fail(h, COREG_ENOTIMPL, "method must be COREG_METHOD_CORRELATION or COREG_METHOD_RESIDUS");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_small, true));
    RETCHK(check_grid(h, grid));
    if (!std::isfinite(solar_r) || !(solar_r > 0.0)) return fail(h, COREG_EINVAL, "solar_r must be positive");
    LagDims d;
    RETCHK(check_lags(h, lags, &d, lag_begin, lag_end, combo));
    RETCHK(bind_device_nowait(h));  // (the image to align is joined right before k_sweep: launch_sweep)
    if (h->ref.p && (h->gW != grid->n_lon || h->gH != grid->n_lat))
        return fail(h, COREG_EINVAL, "reference-on-grid shape differs from the Carrington grid");
    const long long n_out = lag_end - lag_begin;
    double* out_dev = nullptr;
    RETCHK(begin_sweep(h, n_out, corr_out, out_on_device, &out_dev));
    if (n_out == 0) return end_sweep(h, n_out, corr_out, out_on_device, out_dev);  // (nothing to fill)

    CarrDev cd;
    std::memset(&cd, 0, sizeof(cd));
    RETCHK(upload_carr_tables(h, *grid, *hdr_small, &cd));

    // ---- plan: local geometry (heuristic inputs only) -> tile shape + lag patch
    Geometry geo;
    {
        const CarrCommon c0 = carr_common(*hdr_small, solar_r);
        const int ic = grid->n_lon / 2, jc = grid->n_lat / 2;
        double a0, a1, b0, b1, c0x, c0y;
        carr_term_host(h->tabs, c0, ic, jc, &a0, &a1);
        carr_term_host(h->tabs, c0, std::min(ic + 1, grid->n_lon - 1), jc, &b0, &b1);
        carr_term_host(h->tabs, c0, ic, std::min(jc + 1, grid->n_lat - 1), &c0x, &c0y);
        geo.dx_di = b0 - a0;
        geo.dy_di = b1 - a1;
        geo.dx_dj = c0x - a0;
        geo.dy_dj = c0y - a1;
        // utils/rectify.py:399-404: X0 = -(c d1 + s d2)/cdelt1, Y0 = -(-s d1 + c d2)/cdelt2
        const double s1 = lag_step(lags->crval1, d.n1), s2 = lag_step(lags->crval2, d.n2);
        geo.ax = c0.cr * s1 / hdr_small->cdelt1;
        geo.ay = c0.sr * s1 / hdr_small->cdelt2;
        geo.bx = c0.sr * s2 / hdr_small->cdelt1;
        geo.by = c0.cr * s2 / hdr_small->cdelt2;
    }
    const long long row = (long long)d.n2 * d.nc;
    const int m1 = (int)((lag_end - 1) / row) - (int)(lag_begin / row) + 1;
    const Plan plan = choose_plan(h, geo, m1, d.n2, h->opt_use_lds ? lds_window_elems(h) : (1LL << 40));

    PrecomputeArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    {
        const int th = kTilePts / plan.tile_w;
        RETCHK(reserve_tiles(h, ((h->gW + plan.tile_w - 1) / plan.tile_w) * ((h->gH + th - 1) / th)));
    }
    fill_precompute_common(h, &pa, plan.tile_w);
    pa.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
    const int n_tiles = pa.tiles_x * pa.tiles_y;

    // ---- every (cdelt1, cdelt2, crota) combination = one precompute + one sweep launch; all lag parameters of all
    //      launches are staged together and uploaded once
    struct Launch {
        size_t slot_off;
        int n_batches;
        CarrCommon cc;
        double f0lo, f0hi, f1lo, f1hi;
    };
    std::vector<Launch> launches;
    std::vector<double> params;  // per launch: [X0 x ns][Y0 x ns]
    std::vector<long long> outidx;
    SlotList slots;
    for (long long c = 0; c < d.nc; ++c) {
        {
            const long long first = (lag_begin - c + d.nc - 1) / d.nc;  // smallest k with k*nc + c >= begin
            if (first * d.nc + c >= lag_end) continue;
        }
        int i3, i4, i5;
        d.inner(c, &i3, &i4, &i5);
        coreg_wcs2d hc;
        if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[i3], lags->cdelt2[i4], lags->crota[i5], cdelt_semantics,
                         &hc))
            continue;  // reference semantics: this lag kills the worker -> NaN (already filled)
        build_slots(d, c, lag_begin, lag_end, plan.sw, plan.sh, &slots);
        if (slots.n_batches == 0) continue;
        const size_t ns = slots.i1.size();
        Launch L;
        L.slot_off = outidx.size();
        L.n_batches = slots.n_batches;
        L.cc = carr_common(hc, solar_r);
        const size_t pbase = params.size();
        params.resize(pbase + 2 * ns);
        double x0min = 1e300, x0max = -1e300, y0min = 1e300, y0max = -1e300;
        // utils/rectify.py:396-404 with the roll trig hoisted out of the per-lag loop (same values, same order)
        const double roll = hc.crota * kDeg2Rad;
        const double rc = std::cos(roll), rs = std::sin(roll);
        const double nan = std::numeric_limits<double>::quiet_NaN();
        for (size_t s = 0; s < ns; ++s) {
            // padding lanes get NaN: they fail the bounds rule for every point, so waves made only of padding
            // skip every point with one branch (their slots are never written by k_finalize)
            if (slots.outidx[s] < 0) {
                params[pbase + s] = nan;
                params[pbase + ns + s] = nan;
                continue;
            }
            const double v1 = hdr_small->crval1 + lags->crval1[slots.i1[s]];  // alignment.py:404
            const double v2 = hdr_small->crval2 + lags->crval2[slots.i2[s]];  // alignment.py:412
            const double dx = rc * v1 + rs * v2;
            const double dy = -rs * v1 + rc * v2;
            const double x0 = (hc.crpix1 - 1) - dx / hc.cdelt1;
            const double y0 = (hc.crpix2 - 1) - dy / hc.cdelt2;
            params[pbase + s] = x0;
            params[pbase + ns + s] = y0;
            x0min = std::min(x0min, x0);
            x0max = std::max(x0max, x0);
            y0min = std::min(y0min, y0);
            y0max = std::max(y0max, y0);
        }
        outidx.insert(outidx.end(), slots.outidx.begin(), slots.outidx.end());
        // a point can be in bounds for some lag only if X0 + t0 in [0, W-1] for some X0 in [x0min, x0max]
        L.f0lo = -x0max;
        L.f0hi = (double)(h->sW - 1) - x0min;
        L.f1lo = -y0max;
        L.f1hi = (double)(h->sH - 1) - y0min;
        launches.push_back(L);
    }
    if (launches.empty()) {
        RETCHK(fill_nan(h, out_dev, n_out));
        return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
    }
    RETCHK(upload_plan(h, params, outidx, out_dev, n_out));
    RETCHK(prepare_sharded(h, outidx.size(), n_out, lag_begin));
    for (const Launch& L : launches) {
        set_carr_common(&cd, L.cc);
        pa.carr = cd;
        pa.f0lo = L.f0lo;
        pa.f0hi = L.f0hi;
        pa.f1lo = L.f1lo;
        pa.f1hi = L.f1hi;
        {
            // whole-tile skip bound (k_precompute): pixels per radian of grid-point motion, grid steps in radians
            const double dm1 = L.cc.dist - 1.0;
            pa.tile_skip = (h->opt_tile_skip && dm1 > 0.0) ? 1 : 0;
            const double per_rad = pa.tile_skip ? (1.0 / dm1 + 1.0 / (dm1 * dm1)) * kRad2Deg * 3600.0 : 0.0;
            pa.lip_x = per_rad / std::fabs(L.cc.cdelt1) * (1.0 + 1e-9);
            pa.lip_y = per_rad / std::fabs(L.cc.cdelt2) * (1.0 + 1e-9);
            // float32 linspace grid: the spacing is uniform to ~1e-7 relative of the coordinate
            pa.dlon = grid->n_lon > 1 ? (std::fabs(grid->lon1 - grid->lon0) / (grid->n_lon - 1) * 1.001 + 1e-4) * kDeg2Rad : 0.0;
            pa.dlat = grid->n_lat > 1 ? (std::fabs(grid->lat1 - grid->lat0) / (grid->n_lat - 1) * 1.001 + 1e-4) * kDeg2Rad : 0.0;
        }
        {
            const int ng = pick_groups(h, L.n_batches, n_tiles), nb = L.n_batches;
            RETCHK(launch_precompute<MODE_TRANSLATE>(h, pa, n_tiles, ng, nb));
            h->last_precompute = [pa, n_tiles, ng, nb](coreg_handle* hh) {
                return launch_precompute<MODE_TRANSLATE>(hh, pa, n_tiles, ng, nb);
            };
        }
        // SoA block of this launch starts at 2 * slot_off doubles (every earlier launch contributed 2 per slot)
        RETCHK(launch_sweep(h, MODE_TRANSLATE, order, method, h->lane_params.as<double>() + 2 * L.slot_off,
                            h->out_index.as<long long>() + L.slot_off, L.n_batches, n_tiles, lag_begin, out_dev, nullptr,
                            nullptr, (long long)L.slot_off,
                            pick_pitch(h, plan, h->opt_use_lds ? lds_window_elems(h) : 0, order)));
    }
    return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
}

// Plate-carree maps on both sides (Alignment.align_using_initial_carrington, alignment.py:344-399 ->
// _interpolate_on_large_data_grid :1018-1029 with WCS(CRLN-CAR)): the per-lag map is a rotation of the sphere between
// the native frames of the two maps (a CRVAL2 lag makes the shifted map oblique).  One precompute (native angles of
// the target pixels), one sweep launch per (cdelt1, cdelt2, crota) combination (its native -> pixel affine map is a
// launch constant).  Lags whose header has no valid native pole get NaN (astropy raises for them).
static int sweep_car(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small, const coreg_lags* lags,
                     const LagDims& d, int order, int method, int cdelt_semantics, int64_t lag_begin, int64_t lag_end,
                     double* corr_out, int out_on_device, double* out_dev) {
    const long long n_out = lag_end - lag_begin;
    Mat3 r_target;
    if (car_native_to_celestial(*hdr_target, &r_target))
        return fail(h, COREG_EINVAL, "hdr_target: no valid native pole for this CRVAL2 / LONPOLE (CAR)");
    auto shifted_by = [&](const coreg_wcs2d& base, double v1, double v2) {
        coreg_wcs2d hl = base;
        hl.crval1 = hdr_small->crval1 + v1;  // alignment.py:404
        hl.crval2 = hdr_small->crval2 + v2;  // alignment.py:412
        return hl;
    };
    auto shifted = [&](const coreg_wcs2d& base, int i1, int i2) {
        return shifted_by(base, lags->crval1[i1], lags->crval2[i2]);
    };
    // ---- plan: local geometry from the maps of a central lag and of that lag plus one mean step on either axis
    Geometry geo;
    {
        int e1[3], e2[3];
        extreme_lags(lags->crval1, d.n1, e1);
        extreme_lags(lags->crval2, d.n2, e2);
        const double v1 = lags->crval1[e1[1]], v2 = lags->crval2[e2[1]];
        CarMapHost m0, m1h, m2h;
        if (m0.init(*hdr_target, shifted_by(*hdr_small, v1, v2)) ||
            m1h.init(*hdr_target, shifted_by(*hdr_small, v1 + lag_step(lags->crval1, d.n1), v2)) ||
            m2h.init(*hdr_target, shifted_by(*hdr_small, v1, v2 + lag_step(lags->crval2, d.n2)))) {
            geo.dx_di = geo.dy_dj = 1.0;  // central lag invalid: any plan will do, its lanes are NaN
            geo.dy_di = geo.dx_dj = geo.ax = geo.ay = geo.bx = geo.by = 0.0;
        } else {
            const double u = hdr_target->naxis1 * 0.5, v = hdr_target->naxis2 * 0.5;
            double x0, y0, x1, y1;
            m0.apply(u, v, &x0, &y0);
            m0.apply(u + 1, v, &x1, &y1);
            geo.dx_di = x1 - x0;
            geo.dy_di = y1 - y0;
            m0.apply(u, v + 1, &x1, &y1);
            geo.dx_dj = x1 - x0;
            geo.dy_dj = y1 - y0;
            m1h.apply(u, v, &x1, &y1);
            geo.ax = x1 - x0;
            geo.ay = y1 - y0;
            m2h.apply(u, v, &x1, &y1);
            geo.bx = x1 - x0;
            geo.by = y1 - y0;
        }
    }
    const long long row = (long long)d.n2 * d.nc;
    const int m1 = (int)((lag_end - 1) / row) - (int)(lag_begin / row) + 1;
    const Plan plan = choose_plan(h, geo, m1, d.n2, h->opt_use_lds ? lds_window_elems(h) : (1LL << 40));

    // rotation of every (CRVAL1, CRVAL2) lag: R = R_small(lag)^T * R_target  (PC / CDELT do not enter it)
    const int i1_lo = (int)(lag_begin / row), i1_hi = (int)((lag_end - 1) / row);
    std::vector<double> rot((size_t)(i1_hi - i1_lo + 1) * d.n2 * 9);
    const double nanv = std::numeric_limits<double>::quiet_NaN();
    for (int i1 = i1_lo; i1 <= i1_hi; ++i1)
        for (int i2 = 0; i2 < d.n2; ++i2) {
            double* r = &rot[((size_t)(i1 - i1_lo) * d.n2 + i2) * 9];
            Mat3 rs;
            if (car_native_to_celestial(shifted(*hdr_small, i1, i2), &rs)) {
                for (int k = 0; k < 9; ++k) r[k] = nanv;
                continue;
            }
            const Mat3 m = mat_mul(mat_T(rs), r_target);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) r[3 * a + b] = (double)m.m[a][b];
        }

    struct Launch {
        size_t slot_off;
        int n_batches;
        LaunchU inv;
        bool identity;  // the one-slot launch of the identity lag-point (below)
        // the single-sample pass of this launch: the combination's header, the CRVAL lag indices of its
        // slots, and the slots the scan skips -- all but the lags that leave CRVAL1 or CRVAL2 of the target header alone:
        // only those bring whole rows or columns of coordinates (or, with a CROTA / CDELT lag on top, the reference pixel)
        // back within wcslib's noise of integers
        coreg_wcs2d hc;
        std::vector<int> i1, i2;
        std::vector<unsigned char> tap_skip;
        bool tap_any = false;
    };
    std::vector<Launch> launches;
    std::vector<double> params;  // per launch: SoA [9][slots of the launch]
    std::vector<long long> outidx;
    SlotList slots;
    // The identity lag-point (shifted header == target header: the zero lag of the sub-map semantics, where the target IS
    // the header of the map to align).  The reference's pixel -> world -> pixel round trip (alignment.py:1038-1069) returns
    // i + eps there and the sign of wcslib's rounding noise decides the bounds rule on every border pixel (and, for odd
    // spline orders, the tap set of every pixel).  The sphere rotation of this path cannot even return exact integers, so
    // that lag-point is taken out of the CAR launch and swept on its own with the EXACT identity map by the
    // helioprojective kernels, whose zero-lag machinery (k_border_fix / k_parity_fix) then applies what wcslib's chain
    // (geometry.hpp WcslibCar, bit-exact) decides.  "border_fix" 0: rotation path for that lag-point too.
    long long identity_out = -1;
    BorderFix id_fix;
    std::vector<unsigned char> id_flags;
    auto same_header = [](const coreg_wcs2d& a, const coreg_wcs2d& b) {
        auto eq = [](double x, double y) { return x == y || (x != x && y != y); };
        return a.proj == b.proj && a.crpix1 == b.crpix1 && a.crpix2 == b.crpix2 && a.crval1 == b.crval1 &&
               a.crval2 == b.crval2 && a.cdelt1 == b.cdelt1 && a.cdelt2 == b.cdelt2 && a.pc1_1 == b.pc1_1 &&
               a.pc1_2 == b.pc1_2 && a.pc2_1 == b.pc2_1 && a.pc2_2 == b.pc2_2 && a.unit_to_deg == b.unit_to_deg &&
               eq(a.lonpole, b.lonpole) && eq(a.latpole, b.latpole) && a.naxis1 == b.naxis1 && a.naxis2 == b.naxis2;
    };
    for (long long c = 0; c < d.nc; ++c) {
        const long long first = (lag_begin - c + d.nc - 1) / d.nc;
        if (first * d.nc + c >= lag_end) continue;
        int i3, i4, i5;
        d.inner(c, &i3, &i4, &i5);
        coreg_wcs2d hc;
        if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[i3], lags->cdelt2[i4], lags->crota[i5], cdelt_semantics,
                         &hc))
            continue;  // reference semantics: this lag kills the worker -> NaN (already filled)
        build_slots(d, c, lag_begin, lag_end, plan.sw, plan.sh, &slots);
        if (slots.n_batches == 0) continue;
        const size_t ns = slots.i1.size();
        Launch L;
        L.slot_off = outidx.size();
        L.n_batches = slots.n_batches;
        const Affine2 inv = car_native_to_pix(hc);
        std::memset(&L.inv, 0, sizeof(L.inv));
        L.inv.m00 = inv.m00;
        L.inv.m01 = inv.m01;
        L.inv.m10 = inv.m10;
        L.inv.m11 = inv.m11;
        L.inv.b0 = inv.b0;
        L.inv.b1 = inv.b1;
        L.inv.box_c = car_box_c(*hdr_target, hc, plan.tile_w);
        L.inv.pole_sep = 0.0;  // largest over the lags of this launch (below)
        L.identity = false;
        const size_t pbase = params.size();
        params.resize(pbase + 9 * ns);
        for (size_t s = 0; s < ns; ++s) {
            bool pad = slots.outidx[s] < 0;
            if (!pad && h->opt_border_fix && identity_out < 0 && h->gW == h->sW && h->gH == h->sH) {
                const coreg_wcs2d hl = shifted(hc, slots.i1[s], slots.i2[s]);
                if (same_header(hl, *hdr_target)) {
                    identity_out = slots.outidx[s];
                    slots.outidx[s] = -1;  // not this launch's: padding lane (NaN map, nothing written)
                    pad = true;
                }
            }
            const double* r = &rot[((size_t)(slots.i1[s] - i1_lo) * d.n2 + slots.i2[s]) * 9];
            for (int k = 0; k < 9; ++k) params[pbase + (size_t)k * ns + s] = pad ? nanv : r[k];
            if (!pad && r[8] == r[8]) L.inv.pole_sep = std::max(L.inv.pole_sep, car_pole_sep(r));
            unsigned char skip = 1;
            if (!pad && h->opt_tap_fix && r[8] == r[8] && h->gW == h->sW && h->gH == h->sH) {
                // (a CROTA / CDELT lag on top of it included: with both CRVAL equal the map is affine about CRPIX and
                // returns the reference pixel itself to within the noise)
                const coreg_wcs2d hl = shifted(hc, slots.i1[s], slots.i2[s]);
                if (hl.crval1 == hdr_target->crval1 || hl.crval2 == hdr_target->crval2) skip = 0;
            }
            L.tap_skip.push_back(skip);
            L.tap_any = L.tap_any || !skip;
        }
        L.hc = hc;
        L.i1 = slots.i1;
        L.i2 = slots.i2;
        outidx.insert(outidx.end(), slots.outidx.begin(), slots.outidx.end());
        launches.push_back(L);
    }
    if (identity_out >= 0) {
        // one batch, one live slot: the identity homography (every sample ON its pixel); the others are padding
        Launch L;
        std::memset(&L.inv, 0, sizeof(L.inv));
        L.slot_off = outidx.size();
        L.n_batches = 1;
        L.identity = true;
        const size_t ns = kBlock, pbase = params.size();
        params.resize(pbase + 9 * ns, nanv);
        const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k) params[pbase + (size_t)k * ns] = ident[k];
        outidx.push_back(identity_out);
        outidx.insert(outidx.end(), ns - 1, -1);
        launches.push_back(L);
        {   // (on every rank of a grid-sharded sweep: launch_sweep applies it on rank 0 only, but all must agree that this
            // launch carries a correction)
            AxisInvariance inv;
            inv.rows = inv.cols = true;
            BorderFix::Item it;
            it.slot = 0;
            it.first = 0;
            wcslib_dropped_border_pixels(h, *hdr_target, *hdr_target, inv, &id_fix.pixels);
            it.n = (int)id_fix.pixels.size();
            it.flags_off = -1;
            if (order & 1) {
                it.flags_off = 0;
                id_flags = wcslib_tap_shift_flags(h, *hdr_target, *hdr_target, inv);
            }
            if (it.n > 0 || it.flags_off >= 0) id_fix.items.push_back(it);
        }
    }
    if (launches.empty()) {
        RETCHK(fill_nan(h, out_dev, n_out));
        return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
    }
    RETCHK(upload_plan(h, params, outidx, out_dev, n_out));
    RETCHK(prepare_sharded(h, outidx.size(), n_out, lag_begin));

    PrecomputeArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    {
        const int th = kTilePts / plan.tile_w;
        RETCHK(reserve_tiles(h, ((h->gW + plan.tile_w - 1) / plan.tile_w) * ((h->gH + th - 1) / th)));
    }
    fill_precompute_common(h, &pa, plan.tile_w);
    pa.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
    const int n_tiles = pa.tiles_x * pa.tiles_y;
    const Affine2 fwd = car_pix_to_native(*hdr_target);
    pa.car_fwd.m00 = fwd.m00;
    pa.car_fwd.m01 = fwd.m01;
    pa.car_fwd.m10 = fwd.m10;
    pa.car_fwd.m11 = fwd.m11;
    pa.car_fwd.b0 = fwd.b0;
    pa.car_fwd.b1 = fwd.b1;
    const double inf = std::numeric_limits<double>::infinity();
    pa.f0lo = pa.f1lo = -inf;  // no culling by position: only non-finite reference values drop out
    pa.f0hi = pa.f1hi = inf;
    if (!id_fix.items.empty()) RETCHK(upload_border_pixels(h, id_fix.pixels));
    if (!id_flags.empty()) {
        HIPCHK(h->border_flags.reserve(id_flags.size()));
        HIPCHK(hipStreamSynchronize(h->stream));  // (pageable source, rare path)
        HIPCHK(hipMemcpy(h->border_flags.p, id_flags.data(), id_flags.size(), hipMemcpyHostToDevice));
    }
    int last_groups = -1, last_batches = -1;
    for (const Launch& L : launches) {
        if (L.identity) {
            // target pixel -> the same pixel of the map to align: base coordinates = pixel indices, no culling by position
            PrecomputeArgs pi = pa;
            std::memset(&pi.car_fwd, 0, sizeof(pi.car_fwd));
            const int ng = pick_groups(h, 1, n_tiles);
            RETCHK(launch_precompute<MODE_HOMOGRAPHY>(h, pi, n_tiles, ng, 1));
            h->last_precompute = [pi, n_tiles, ng](coreg_handle* hh) {
                return launch_precompute<MODE_HOMOGRAPHY>(hh, pi, n_tiles, ng, 1);
            };
            last_groups = last_batches = -1;  // (the compacted points now hold pixel indices, not unit vectors)
            RETCHK(launch_sweep(h, MODE_HOMOGRAPHY, order, method, h->lane_params.as<double>() + 9 * L.slot_off,
                                h->out_index.as<long long>() + L.slot_off, 1, n_tiles, lag_begin, out_dev, nullptr, &id_fix,
                                (long long)L.slot_off));
            continue;
        }
        // the work partition (k_tile_list) depends on the group count of the launch: redo it only when that changes
        const int ng = pick_groups(h, L.n_batches, n_tiles);
        if (ng != last_groups || L.n_batches != last_batches) RETCHK(launch_precompute<MODE_CAR>(h, pa, n_tiles, ng, L.n_batches));
        {
            const int nb = L.n_batches;
            h->last_precompute = [pa, n_tiles, ng, nb](coreg_handle* hh) {
                return launch_precompute<MODE_CAR>(hh, pa, n_tiles, ng, nb);
            };
        }
        last_groups = ng;
        last_batches = L.n_batches;
        BorderFix tap;  // (no whole-grid items here: the identity lag has its own launch)
        if (L.tap_any) {
            const double box[4] = {0.0, (double)(h->gW - 1), 0.0, (double)(h->gH - 1)};
            h->tap_last[0] = h->tap_last[1] = h->tap_last[2] = 0;
            RETCHK(prepare_tap_fix(
                h, MODE_CAR, order, *hdr_target, (long long)L.tap_skip.size(), L.tap_skip, box,
                [&](int slot) { return shifted(L.hc, L.i1[(size_t)slot], L.i2[(size_t)slot]); }, &tap,
                h->lane_params.as<double>() + 9 * L.slot_off, &L.inv, &pa.car_fwd));
        }
        RETCHK(launch_sweep(h, MODE_CAR, order, method, h->lane_params.as<double>() + 9 * L.slot_off,
                            h->out_index.as<long long>() + L.slot_off, L.n_batches, n_tiles, lag_begin, out_dev,
                            &L.inv, L.tap_any ? &tap : nullptr, (long long)L.slot_off));
    }
    return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
}

int coreg_sweep_helioprojective(coreg_handle* h, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small,
                                const coreg_lags* lags, int order, int method, int cdelt_semantics, int64_t lag_begin,
                                int64_t lag_end, double* corr_out, int out_on_device) {
    if (!h) return COREG_EINVAL;
    const ComboRange combo = take_combo_range(h);
    if (!hdr_target || !hdr_small) return fail(h, COREG_EINVAL, "sweep_helioprojective: null header");
    if (hdr_target->proj != hdr_small->proj || (hdr_small->proj != COREG_PROJ_TAN && hdr_small->proj != COREG_PROJ_CAR))
        return fail(h, COREG_ENOTIMPL, "both headers must be TAN (helioprojective) or both CAR (Carrington maps)");
    if (method != COREG_METHOD_CORRELATION && method != COREG_METHOD_RESIDUS)
        return fail(h, COREG_ENOTIMPL, "method must be COREG_METHOD_CORRELATION or COREG_METHOD_RESIDUS");
    RETCHK(check_order(h, order));
    RETCHK(check_wcs(h, hdr_target, false));
    RETCHK(check_wcs(h, hdr_small, false));
    LagDims d;
    RETCHK(check_lags(h, lags, &d, lag_begin, lag_end, combo));
    RETCHK(bind_device_nowait(h));
    if (h->ref.p && (h->gW != hdr_target->naxis1 || h->gH != hdr_target->naxis2))
        return fail(h, COREG_EINVAL, "reference-on-grid shape differs from hdr_target NAXIS1/NAXIS2");
    const long long n_out = lag_end - lag_begin;
    double* out_dev = nullptr;
    RETCHK(begin_sweep(h, n_out, corr_out, out_on_device, &out_dev));
    if (n_out == 0) return end_sweep(h, n_out, corr_out, out_on_device, out_dev);  // (nothing to fill)
    if (hdr_small->proj == COREG_PROJ_CAR)
        return sweep_car(h, hdr_target, hdr_small, lags, d, order, method, cdelt_semantics, lag_begin, lag_end, corr_out,
                         out_on_device, out_dev);

    // ---- plan: local geometry from the maps of the central lag and of its two neighbours
    Geometry geo;
    {
        auto map_of = [&](double v1, double v2, double hm[9]) {
            coreg_wcs2d hl = *hdr_small;
            hl.crval1 = hdr_small->crval1 + v1;
            hl.crval2 = hdr_small->crval2 + v2;
            homography(*hdr_target, hl, hm);
        };
        int e1[3], e2[3];
        extreme_lags(lags->crval1, d.n1, e1);
        extreme_lags(lags->crval2, d.n2, e2);
        const double v1 = lags->crval1[e1[1]], v2 = lags->crval2[e2[1]];
        double m0[9], m1h[9], m2h[9];
        map_of(v1, v2, m0);
        map_of(v1 + lag_step(lags->crval1, d.n1), v2, m1h);
        map_of(v1, v2 + lag_step(lags->crval2, d.n2), m2h);
        const double u = hdr_target->naxis1 * 0.5, v = hdr_target->naxis2 * 0.5;
        double x0, y0, x1, y1;
        apply_h(m0, u, v, &x0, &y0);
        apply_h(m0, u + 1, v, &x1, &y1);
        geo.dx_di = x1 - x0;
        geo.dy_di = y1 - y0;
        apply_h(m0, u, v + 1, &x1, &y1);
        geo.dx_dj = x1 - x0;
        geo.dy_dj = y1 - y0;
        apply_h(m1h, u, v, &x1, &y1);
        geo.ax = x1 - x0;
        geo.ay = y1 - y0;
        apply_h(m2h, u, v, &x1, &y1);
        geo.bx = x1 - x0;
        geo.by = y1 - y0;
    }
    const long long row = (long long)d.n2 * d.nc;
    const int m1 = (int)((lag_end - 1) / row) - (int)(lag_begin / row) + 1;
    const Plan plan = choose_plan(h, geo, m1, d.n2, h->opt_use_lds ? lds_window_elems(h) : (1LL << 40));

    // ---- all slots of all (cdelt1, cdelt2, crota) combinations -> ONE launch
    // Two passes.  (1) per combination, independent of every other and of the handle -- a few host threads share them
    // when the lag set is large (cfg4: 21 combinations x 3 721 lag-points, 1.4 ms of 3 x 3 products on one core):
    // shifted header, slots, one homography per slot, the combination's corner of the cull box.  (2) in combination
    // order, on this thread: the lag-points decided by wcslib's rounding noise (handle caches), the bookkeeping of the
    // odd-order pass, the concatenation.
    std::vector<long long> outidx;
    int n_batches = 0;
    double fx0 = 1e300, fx1 = -1e300, fy0 = 1e300, fy1 = -1e300;  // cull box in target pixels
    HomographyFamily fam;
    fam.init(*hdr_target, *hdr_small, lags->crval1, d.n1, lags->crval2, d.n2, d.nc > 1);
    BorderFix fix;
    std::vector<std::vector<unsigned char>> flags_host;  // per noise-decided lag-point (odd spline orders only)
    // odd spline orders: what prepare_tap_fix needs to rebuild a slot's shifted header
    // (grid shares across GPUs: every rank lists the samples -- the re-evaluation of a flagged lag-point needs them on
    // every rank; the correction itself is launched on rank 0 only, launch_sweep)
    // Odd orders: every sample within 1e-8 px of an integer coordinate (the sign of wcslib's noise picks the taps); even
    // orders: only those within 1e-8 px of a BOUND of the image (the sign decides the bounds rule) -- a pure CRVAL1 or
    // CRVAL2 lag under an unrotated header keeps whole border rows / columns of the grid there.
    const bool tap_fixing = h->opt_tap_fix != 0;
    std::vector<coreg_wcs2d> tap_combo;      // the (cdelt, crota)-shifted header of each combination
    std::vector<int> tap_slot_combo, tap_slot_i1, tap_slot_i2;
    std::vector<unsigned char> tap_skip;     // padding lanes and lag-points the structured fix handles
    const int i1_lo = (int)(lag_begin / row), i1_hi = (int)((lag_end - 1) / row);
    fam.fill_products(i1_lo, i1_hi);  // (read-only from here on: `get` is safe to call from several threads)
    const double nanv = std::numeric_limits<double>::quiet_NaN();
    struct ComboPlan {
        bool used = false;
        coreg_wcs2d hc;
        SlotList slots;
        std::vector<double> hs;  // AoS [slot][9]
        double box[4] = {1e300, -1e300, 1e300, -1e300};
    };
    std::vector<ComboPlan> cps((size_t)d.nc);
    // which target pixels can ever be in bounds: inverse maps of the small image's corners for the lags on the
    // boundary of the (CRVAL1, CRVAL2) rectangle, chosen BY VALUE (the reference accepts lag lists in any order):
    // smallest, largest and the value nearest the middle of each axis (the maps vary smoothly and monotonically
    // with the lag value, the +-3 px margin below covers the curvature in between)
    int e1[3], e2[3];
    extreme_lags(lags->crval1 + i1_lo, i1_hi - i1_lo + 1, e1);
    extreme_lags(lags->crval2, d.n2, e2);
    auto plan_combo = [&](long long c) {
        ComboPlan& cp = cps[(size_t)c];
        const long long first = (lag_begin - c + d.nc - 1) / d.nc;
        if (first * d.nc + c >= lag_end) return;
        int i3, i4, i5;
        d.inner(c, &i3, &i4, &i5);
        if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[i3], lags->cdelt2[i4], lags->crota[i5], cdelt_semantics,
                         &cp.hc))
            return;
        build_slots(d, c, lag_begin, lag_end, plan.sw, plan.sh, &cp.slots);
        if (cp.slots.n_batches == 0) return;
        cp.used = true;
        const Mat3d B = HomographyFamily::combo(cp.hc);
        const size_t n = cp.slots.i1.size();
        cp.hs.resize(9 * n);
        for (size_t s = 0; s < n; ++s) {
            double* hm = &cp.hs[9 * s];
            if (cp.slots.outidx[s] < 0) {  // padding lane: NaN map -> never in bounds
                for (int k = 0; k < 9; ++k) hm[k] = nanv;
            } else {
                fam.get(B, cp.slots.i1[s], cp.slots.i2[s], hm);
            }
        }
        for (int a1 = 0; a1 < 3; ++a1)
            for (int a2 = 0; a2 < 3; ++a2) {
                coreg_wcs2d hl = cp.hc;
                hl.crval1 = hdr_small->crval1 + lags->crval1[i1_lo + e1[a1]];
                hl.crval2 = hdr_small->crval2 + lags->crval2[e2[a2]];
                double hi[9];
                homography(hl, *hdr_target, hi);
                for (int k = 0; k < 4; ++k) {
                    double px, py;
                    apply_h(hi, (k & 1) ? (double)(h->sW - 1) : 0.0, (k & 2) ? (double)(h->sH - 1) : 0.0, &px, &py);
                    cp.box[0] = std::min(cp.box[0], px);
                    cp.box[1] = std::max(cp.box[1], px);
                    cp.box[2] = std::min(cp.box[2], py);
                    cp.box[3] = std::max(cp.box[3], py);
                }
            }
    };
    const unsigned plan_threads = (d.nc >= 2 && (long long)d.nc * d.n1 * d.n2 >= 16384)
                                      ? std::min<unsigned>({8u, (unsigned)d.nc, std::max(1u, std::thread::hardware_concurrency())})
                                      : 1u;
    if (plan_threads <= 1) {
        for (long long c = 0; c < d.nc; ++c) plan_combo(c);
    } else {
        std::atomic<long long> next(0);
        auto worker = [&] {
            for (long long c = next.fetch_add(1); c < d.nc; c = next.fetch_add(1)) plan_combo(c);
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < plan_threads; ++t) th.emplace_back(worker);
        worker();
        for (auto& x : th) x.join();
    }
    for (long long c = 0; c < d.nc; ++c) {
        ComboPlan& cp = cps[(size_t)c];
        if (!cp.used) continue;
        const SlotList& slots = cp.slots;
        const coreg_wcs2d& hc = cp.hc;
        if (tap_fixing) {
            tap_combo.push_back(hc);
            for (size_t s = 0; s < slots.i1.size(); ++s) {
                tap_slot_combo.push_back((int)tap_combo.size() - 1);
                tap_slot_i1.push_back(slots.outidx[s] < 0 ? 0 : slots.i1[s]);
                tap_slot_i2.push_back(slots.outidx[s] < 0 ? 0 : slots.i2[s]);
                tap_skip.push_back(slots.outidx[s] < 0 ? 1 : 0);
            }
        }
        if (h->opt_border_fix) {
            for (size_t s = 0; s < slots.i1.size(); ++s) {
                if (slots.outidx[s] < 0) continue;
                // same tangent point as the target (sub-map path, zero CRVAL lag) and an invariant image axis: exact
                // invariant map on the device, border pixels decided as the reference's wcslib round trip decides them
                // (geometry.hpp WcslibTan, k_border_fix).  (The tangent points are compared first: every other
                // lag-point is dismissed without building its header.)
                const double v1 = hdr_small->crval1 + lags->crval1[slots.i1[s]];
                const double v2 = hdr_small->crval2 + lags->crval2[slots.i2[s]];
                if (v1 != hdr_target->crval1 || v2 != hdr_target->crval2) continue;
                coreg_wcs2d hl = hc;
                hl.crval1 = v1;
                hl.crval2 = v2;
                double* hm = &cp.hs[9 * s];
                const AxisInvariance inv = snap_invariant_axes(*hdr_target, hl, h->gW, h->gH, hm);
                if (inv.rows || inv.cols) {
                    BorderFix::Item it;
                    it.slot = (long long)(outidx.size() + s);
                    it.first = (int)fix.pixels.size();
                    wcslib_dropped_border_pixels(h, *hdr_target, hl, inv, &fix.pixels);
                    it.n = (int)fix.pixels.size() - it.first;
                    it.flags_off = -1;
                    if (order & 1) {
                        it.flags_off = (long long)flags_host.size() * h->gW * h->gH;
                        flags_host.push_back(wcslib_tap_shift_flags(h, *hdr_target, hl, inv));  // (copy: the cache may evict)
                    }
                    if (it.n > 0 || it.flags_off >= 0) fix.items.push_back(it);
                    if (tap_fixing) tap_skip[(size_t)it.slot] = 1;  // (its whole grid sits on integers: k_parity_fix)
                }
            }
        }
        fx0 = std::min(fx0, cp.box[0]);
        fx1 = std::max(fx1, cp.box[1]);
        fy0 = std::min(fy0, cp.box[2]);
        fy1 = std::max(fy1, cp.box[3]);
        outidx.insert(outidx.end(), slots.outidx.begin(), slots.outidx.end());
        n_batches += slots.n_batches;
    }
    if (n_batches == 0) {
        RETCHK(fill_nan(h, out_dev, n_out));
        return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
    }
    const size_t ns = outidx.size();
    std::vector<double> params(9 * ns);
    double eps_max = 0.0;  // largest |h6 x + h7 y| over the target grid and all lags
    {
        // AoS per combination -> SoA [9][ns] of the launch, the same threads over the combinations
        std::vector<size_t> off((size_t)d.nc + 1, 0);
        for (long long c = 0; c < d.nc; ++c) off[(size_t)c + 1] = off[(size_t)c] + (cps[(size_t)c].used ? cps[(size_t)c].slots.i1.size() : 0);
        std::vector<double> eps_of((size_t)d.nc, 0.0);
        auto transpose = [&](long long c) {
            const ComboPlan& cp = cps[(size_t)c];
            if (!cp.used) return;
            const size_t n = cp.slots.i1.size(), at = off[(size_t)c];
            double em = 0.0;
            for (size_t s = 0; s < n; ++s) {
                const double* hm = &cp.hs[9 * s];
                for (int k = 0; k < 9; ++k) params[(size_t)k * ns + at + s] = hm[k];
                const double e = std::fabs(hm[6]) * (double)h->gW + std::fabs(hm[7]) * (double)h->gH;
                if (e == e) em = std::max(em, e);
            }
            eps_of[(size_t)c] = em;
        };
        if (plan_threads <= 1) {
            for (long long c = 0; c < d.nc; ++c) transpose(c);
        } else {
            std::atomic<long long> next(0);
            auto worker = [&] {
                for (long long c = next.fetch_add(1); c < d.nc; c = next.fetch_add(1)) transpose(c);
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < plan_threads; ++t) th.emplace_back(worker);
            worker();
            for (auto& x : th) x.join();
        }
        for (double e : eps_of) eps_max = std::max(eps_max, e);
    }
    // 1/(1 + eps) = 1 - eps + eps^2 is exact to float64 below ~4e-6 (eps^3 < 1e-16); wider fields divide exactly
    const int sweep_mode = (h->opt_h_series && eps_max < 4.0e-6) ? MODE_HOMOGRAPHY_SERIES : MODE_HOMOGRAPHY;
    RETCHK(upload_plan(h, params, outidx, out_dev, n_out));
    RETCHK(prepare_sharded(h, outidx.size(), n_out, lag_begin));

    PrecomputeArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    {
        const int th = kTilePts / plan.tile_w;
        RETCHK(reserve_tiles(h, ((h->gW + plan.tile_w - 1) / plan.tile_w) * ((h->gH + th - 1) / th)));
    }
    fill_precompute_common(h, &pa, plan.tile_w);
    pa.residus = method == COREG_METHOD_RESIDUS ? 1 : 0;
    const int n_tiles = pa.tiles_x * pa.tiles_y;
    // the maps are projective and the image corners bound its interior
    pa.f0lo = std::floor(fx0) - 3.0;
    pa.f0hi = std::ceil(fx1) + 3.0;
    pa.f1lo = std::floor(fy0) - 3.0;
    pa.f1hi = std::ceil(fy1) + 3.0;
    if (!fix.items.empty()) RETCHK(upload_border_pixels(h, fix.pixels));
    if (!flags_host.empty()) {
        const size_t each = (size_t)h->gW * h->gH;
        HIPCHK(h->border_flags.reserve(each * flags_host.size()));
        HIPCHK(hipStreamSynchronize(h->stream));  // (pageable source, rare path: blocking copies are fine)
        for (size_t k = 0; k < flags_host.size(); ++k)
            HIPCHK(hipMemcpy(h->border_flags.as<unsigned char>() + k * each, flags_host[k].data(), each,
                             hipMemcpyHostToDevice));
    }
    {
        const int ng = pick_groups(h, n_batches, n_tiles);
        RETCHK(launch_precompute<MODE_HOMOGRAPHY>(h, pa, n_tiles, ng, n_batches));
        h->last_precompute = [pa, n_tiles, ng, n_batches](coreg_handle* hh) {
            return launch_precompute<MODE_HOMOGRAPHY>(hh, pa, n_tiles, ng, n_batches);
        };
    }
    h->tap_last[0] = h->tap_last[1] = h->tap_last[2] = 0;
    const double tap_box[4] = {pa.f0lo, pa.f0hi, pa.f1lo, pa.f1hi};
    if (tap_fixing)
        RETCHK(prepare_tap_fix(
            h, sweep_mode, order, *hdr_target, (long long)ns, tap_skip, tap_box,
            [&](int slot) {
                coreg_wcs2d hl = tap_combo[(size_t)tap_slot_combo[(size_t)slot]];
                hl.crval1 = hdr_small->crval1 + lags->crval1[tap_slot_i1[(size_t)slot]];
                hl.crval2 = hdr_small->crval2 + lags->crval2[tap_slot_i2[(size_t)slot]];
                return hl;
            },
            &fix));
    RETCHK(launch_sweep(h, sweep_mode, order, method, h->lane_params.as<double>(), h->out_index.as<long long>(), n_batches,
                        n_tiles, lag_begin, out_dev, nullptr, &fix, 0,
                        pick_pitch(h, plan, h->opt_use_lds ? lds_window_elems(h) : 0, order)));
    return end_sweep(h, n_out, corr_out, out_on_device, out_dev);
}

int coreg_sums_size(coreg_handle* h, int64_t* n_doubles) {
    if (!h || !n_doubles) return COREG_EINVAL;
    *n_doubles = (int64_t)h->sums_slots * kNumSums;
    return COREG_OK;
}

int coreg_copy_sums(coreg_handle* h, double* dst, int dst_on_device) {
    if (!h || !dst) return COREG_EINVAL;
    if (h->sums_slots <= 0 || h->pending_fin.empty()) return fail(h, COREG_ESTATE, "no point-sharded sweep is pending");
    RETCHK(bind_device(h));
    const size_t bytes = (size_t)h->sums_slots * kNumSums * sizeof(double);
    HIPCHK(hipMemcpyAsync(dst, h->sums.p, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          h->stream));
    if (!dst_on_device) HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_finalize_sums(coreg_handle* h, const double* sums, int sums_on_device, double* corr_out, int out_on_device) {
    if (!h || !sums) return COREG_EINVAL;
    if (h->sums_slots <= 0 || h->pending_fin.empty()) return fail(h, COREG_ESTATE, "no point-sharded sweep is pending");
    if (!corr_out && h->pending_n_out > 0) return fail(h, COREG_EINVAL, "corr_out is null");
    RETCHK(bind_device(h));
    const long long n_out = h->pending_n_out;
    const size_t bytes = (size_t)h->sums_slots * kNumSums * sizeof(double);
    if ((const void*)sums != h->sums.p) {
        HIPCHK(hipMemcpyAsync(h->sums.p, sums, bytes, sums_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              h->stream));
        // a host buffer belongs to the caller again on return (with a device destination nothing below waits)
        if (!sums_on_device) HIPCHK(hipStreamSynchronize(h->stream));
    }
    double* out_dev = corr_out;
    if (!out_on_device) {
        HIPCHK(h->out_dev.reserve((size_t)std::max<long long>(n_out, 1) * sizeof(double)));
        out_dev = h->out_dev.as<double>();
    }
    if (n_out > 0) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, h->stream, out_dev,
                           (long long)n_out, std::numeric_limits<double>::quiet_NaN());
        HIPCHK(hipGetLastError());
    }
    const size_t n_pending = h->pending_fin.size();
    size_t points_of = n_pending - 1;  // the launch whose compacted points the handle holds: the sweep's last one
    for (size_t ip = 0; ip < n_pending; ++ip) {
        const coreg_handle::PendingFinalize& pf = h->pending_fin[ip];
        FinalizeArgs f = {};
        // flags from the REDUCED sums: the same on every rank.  (work-space pointers taken afresh: a later launch of the
        // sweep may have grown the buffers)
        RETCHK(fill_refine(h, &f.refine, pf.refine.mode, pf.refine.order, pf.refine.lane_params, pf.refine.car_inv,
                           pf.n_slots));
        f.refine.enabled = pf.refine.enabled;
        const RefineArgs rf = f.refine;
        f.refine_count = h->counters.as<long long>();
        f.partials = h->sums.as<double>() + pf.slot_off;
        f.n_groups = 1;
        f.n_slots = pf.n_slots;
        f.part_stride = h->sums_slots;
        f.out_index = h->fin_outidx.as<long long>() + pf.slot_off;
        f.lag_begin = pf.lag_begin;
        f.out = out_dev;
        f.residus = pf.residus;
        f.n_required = (long long)h->gW * h->gH;
        f.sums_out = nullptr;
        f.sums_stride = f.sums_off = 0;
        hipLaunchKernelGGL(k_finalize, dim3((unsigned)((pf.n_slots + kFinSlots - 1) / kFinSlots)), dim3(kFinSlots * kFinLanes),
                           0, h->stream, f);
        if (!rf.enabled) continue;
        // Ill-conditioned lag-points: every rank holds both images and re-evaluates them over the WHOLE grid (not its
        // share) with the same kernels in the same order -- identical coefficients on every rank, and equal to the
        // single-GPU sweep's, without a second collective.  The compacted points of a launch that was not the sweep's
        // last have been overwritten by the later launches: computed again, only when something is flagged.
        hipLaunchKernelGGL(k_refine_list, dim3(1), dim3(kListThreads), 0, h->stream, rf, pf.n_slots, h->counters.as<long long>());
        HIPCHK(hipGetLastError());
        int head[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(head, rf.head, sizeof(head), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (head[0] == 0) continue;
        if (points_of != ip && pf.replay_precompute) {
            RETCHK(pf.replay_precompute(h));
            points_of = ip;
        }
        RefineArgs rf2 = rf;
        if (!pf.fixes.empty()) {
            // the launch's noise-decided samples about the flagged slots' own pivots (as launch_sweep does on one GPU)
            HIPCHK(h->rf_fix_slab.reserve((size_t)kNumSums * pf.n_slots * sizeof(double)));
            HIPCHK(hipMemsetAsync(h->rf_fix_slab.p, 0, (size_t)kNumSums * pf.n_slots * sizeof(double), h->stream));
            rf2.fix_slab = h->rf_fix_slab.as<double>();
            RETCHK(launch_fix_kernels(h, pf.fixes, h->rf_fix_slab.as<double>(), rf.slot_pivots, rf.flags));
        }
        RETCHK(launch_refine(h, rf2, pf.n_slots, f.out_index, pf.lag_begin, out_dev, false));
    }
    HIPCHK(hipGetLastError());
    if (!out_on_device && n_out > 0) {
        HIPCHK(hipMemcpyAsync(corr_out, out_dev, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return COREG_OK;
}

int coreg_get_pivots(coreg_handle* h, double* pivots2) {
    if (!h || !pivots2) return COREG_EINVAL;
    RETCHK(bind_device(h));
    HIPCHK(hipMemcpyAsync(pivots2, h->pivots.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return COREG_OK;
}

int coreg_set_pivots(coreg_handle* h, const double* pivots2) {
    if (!h || !pivots2) return COREG_EINVAL;
    if (!(pivots2[0] == pivots2[0]) || !(pivots2[1] == pivots2[1]) || std::isinf(pivots2[0]) || std::isinf(pivots2[1]))
        return fail(h, COREG_EINVAL, "set_pivots: pivots must be finite");
    RETCHK(bind_device(h));
    HIPCHK(hipMemcpyAsync(h->pivots.p, pivots2, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // the caller's two doubles are free again on return
    return COREG_OK;
}

int coreg_last_stats(coreg_handle* h, coreg_stats* out) {
    if (!h || !out) return COREG_EINVAL;
    RETCHK(bind_device(h));
    RETCHK(collect_stats(h));  // waits for an in-flight device-output sweep
    *out = h->stats;
    return COREG_OK;
}

// ---- host-only helpers (no GPU needed): exported so that the header logic can be tested on CPU --------------------
int coreg_shift_header(const coreg_wcs2d* ref, double d_crval1, double d_crval2, double d_cdelt1, double d_cdelt2,
                       double d_crota, int cdelt_semantics, coreg_wcs2d* out) {
    if (!ref || !out) return COREG_EINVAL;
    return shift_header(*ref, d_crval1, d_crval2, d_cdelt1, d_cdelt2, d_crota, cdelt_semantics, out);
}

int coreg_car_map(const coreg_wcs2d* from, const coreg_wcs2d* to, int64_t n, const double* px, const double* py,
                  double* ox, double* oy) {
    if (!from || !to || n < 0 || (n > 0 && (!px || !py || !ox || !oy))) return COREG_EINVAL;
    if (from->proj != COREG_PROJ_CAR || to->proj != COREG_PROJ_CAR) return COREG_EINVAL;
    CarMapHost m;
    if (m.init(*from, *to)) return 1;
    for (int64_t i = 0; i < n; ++i) m.apply(px[i], py[i], &ox[i], &oy[i]);
    return COREG_OK;
}

int coreg_last_visit_counts(coreg_handle* h, int64_t* counts6) {
    int64_t* counts5 = counts6;
    if (!h || !counts5) return COREG_EINVAL;
    RETCHK(bind_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    long long info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (h->tile_info.p) HIPCHK(hipMemcpy(info, h->tile_info.p, 7 * sizeof(long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) counts5[k] = info[3 + k];
    long long refined[2] = {0, 0};
    if (h->counters.p) HIPCHK(hipMemcpy(refined, h->counters.p, sizeof(refined), hipMemcpyDeviceToHost));
    counts5[4] = refined[0];
    counts6[5] = refined[1];
    return COREG_OK;
}

int coreg_last_tap_fix(coreg_handle* h, int64_t* counts3) {
    if (!h || !counts3) return COREG_EINVAL;
    for (int k = 0; k < 3; ++k) counts3[k] = h->tap_last[k];
    return COREG_OK;
}

int coreg_car_tile_margin(const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_shifted, int32_t tile_w,
                          double tile_abs_lat_rad, double* margin_px) {
    if (!hdr_target || !hdr_shifted || !margin_px || tile_w < 1 || tile_w > kTilePts) return COREG_EINVAL;
    if (hdr_target->proj != COREG_PROJ_CAR || hdr_shifted->proj != COREG_PROJ_CAR) return COREG_EINVAL;
    CarMapHost m;
    if (m.init(*hdr_target, *hdr_shifted)) return 1;
    *margin_px = car_tile_margin(car_box_c(*hdr_target, *hdr_shifted, tile_w), tile_abs_lat_rad + car_pole_sep(m.r));
    return COREG_OK;
}

int coreg_homography(const coreg_wcs2d* from, const coreg_wcs2d* to, double* h9) {
    if (!from || !to || !h9) return COREG_EINVAL;
    if (from->proj != COREG_PROJ_TAN || to->proj != COREG_PROJ_TAN) return COREG_ENOTIMPL;
    homography(*from, *to, h9);
    return COREG_OK;
}

int coreg_lag_homography(const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small, const coreg_lags* lags,
                         const int32_t idx[5], int cdelt_semantics, double* h9) {
    if (!hdr_target || !hdr_small || !lags || !idx || !h9) return COREG_EINVAL;
    if (hdr_target->proj != COREG_PROJ_TAN || hdr_small->proj != COREG_PROJ_TAN) return COREG_ENOTIMPL;
    if (idx[0] < 0 || idx[0] >= lags->n_crval1 || idx[1] < 0 || idx[1] >= lags->n_crval2 || idx[2] < 0 ||
        idx[2] >= lags->n_cdelt1 || idx[3] < 0 || idx[3] >= lags->n_cdelt2 || idx[4] < 0 || idx[4] >= lags->n_crota)
        return COREG_EINVAL;
    coreg_wcs2d hc;
    if (shift_header(*hdr_small, 0.0, 0.0, lags->cdelt1[idx[2]], lags->cdelt2[idx[3]], lags->crota[idx[4]],
                     cdelt_semantics, &hc))
        return 1;
    HomographyFamily fam;
    fam.init(*hdr_target, *hdr_small, lags->crval1, lags->n_crval1, lags->crval2, lags->n_crval2);
    fam.get(HomographyFamily::combo(hc), idx[0], idx[1], h9);
    return COREG_OK;
}

int coreg_wcslib_pixel_to_pixel(const coreg_wcs2d* from, const coreg_wcs2d* to, int64_t n, const double* px,
                                const double* py, double* ox, double* oy, double* lng, double* lat) {
    if (!from || !to || n < 0 || (n > 0 && (!px || !py || !ox || !oy))) return COREG_EINVAL;
    if (from->proj == COREG_PROJ_CAR && to->proj == COREG_PROJ_CAR) {
        WcslibCar a, b;
        a.init(*from);
        b.init(*to);
        if (!a.valid || !b.valid) return COREG_EINVAL;
        for (int64_t i = 0; i < n; ++i) {
            double l, t;
            a.p2s(px[i], py[i], &l, &t);
            if (lng) lng[i] = l;
            if (lat) lat[i] = t;
            b.s2p(l, t, &ox[i], &oy[i]);
        }
        return COREG_OK;
    }
    if (from->proj != COREG_PROJ_TAN || to->proj != COREG_PROJ_TAN) return COREG_ENOTIMPL;
    WcslibTan a, b;
    a.init(*from);
    b.init(*to);
    for (int64_t i = 0; i < n; ++i) {
        double l, t;
        a.p2s(px[i], py[i], &l, &t);
        if (lng) lng[i] = l;
        if (lat) lat[i] = t;
        b.s2p(ang2pipi_deg(l), ang2pipi_deg(t), &ox[i], &oy[i]);
    }
    return COREG_OK;
}

int coreg_carrington_origin(const coreg_wcs2d* hdr, double* x0, double* y0) {
    if (!hdr || !x0 || !y0) return COREG_EINVAL;
    carr_origin(*hdr, x0, y0);
    return COREG_OK;
}

int coreg_nansum_planes_be(const void* cube, int32_t bitpix, int64_t n_pixels, const int64_t* plane_index, int32_t n_sel,
                           double* out) {
    if (!cube || !out || n_pixels < 0 || n_sel < 0 || (n_sel > 0 && !plane_index) || (bitpix != -32 && bitpix != -64))
        return COREG_EINVAL;
    for (int k = 0; k < n_sel; ++k)
        if (plane_index[k] < 0) return COREG_EINVAL;
    auto work = [&](int64_t lo, int64_t hi) {
        for (int64_t p = lo; p < hi; ++p) out[p] = 0.0;
        for (int k = 0; k < n_sel; ++k) {
            if (bitpix == -32) {
                const uint32_t* src = (const uint32_t*)cube + (size_t)plane_index[k] * (size_t)n_pixels;
                for (int64_t p = lo; p < hi; ++p) {
                    const uint32_t u = __builtin_bswap32(src[p]);
                    float f;
                    std::memcpy(&f, &u, sizeof(f));
                    const double v = (double)f;
                    out[p] += (v != v) ? 0.0 : v;
                }
            } else {
                const uint64_t* src = (const uint64_t*)cube + (size_t)plane_index[k] * (size_t)n_pixels;
                for (int64_t p = lo; p < hi; ++p) {
                    const uint64_t u = __builtin_bswap64(src[p]);
                    double v;
                    std::memcpy(&v, &u, sizeof(v));
                    out[p] += (v != v) ? 0.0 : v;
                }
            }
        }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int nt = (int)std::min<int64_t>(std::min<unsigned>(hw, 12u), std::max<int64_t>(1, n_pixels * std::max(n_sel, 1) / (1 << 18)));
    if (nt <= 1) {
        work(0, n_pixels);
        return COREG_OK;
    }
    std::vector<std::thread> th;
    const int64_t per = ((n_pixels + nt - 1) / nt + 7) & ~(int64_t)7;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, std::min<int64_t>(n_pixels, t * per), std::min<int64_t>(n_pixels, (t + 1) * per));
    work(0, std::min<int64_t>(n_pixels, per));
    for (auto& t : th) t.join();
    return COREG_OK;
}

int coreg_fit_gaussian2d(int32_t m, const double* x, const double* y, const double* z, const double* p0,
                         const double* lb, const double* ub, int32_t jac, double ftol, double xtol, double gtol,
                         int32_t max_nfev, double* popt, int32_t* nfev, int32_t* status) {
    if (!x || !y || !z || !p0 || !lb || !ub || !popt || !status || m < 1 || m > coregfit::MMAX) return COREG_EINVAL;
    coregfit::Problem P{m, x, y, z};
    int n = 0;
    const int st = coregfit::fit(P, p0, lb, ub, jac != 0, ftol > 0 ? ftol : 1e-8, xtol > 0 ? xtol : 1e-8,
                                 gtol > 0 ? gtol : 1e-8, max_nfev, popt, &n, nullptr);
    if (st == -2) return COREG_EINVAL;
    if (nfev) *nfev = n;
    *status = st;
    return COREG_OK;
}

}  // extern "C"

#include "multi.hpp"
