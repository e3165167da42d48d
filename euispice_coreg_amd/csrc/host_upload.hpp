// Part of libcoreg_hip.so's ONE translation unit (coreg_hip.hip includes the parts in order; round 6 split by concern,
// no behaviour change): images to the device: staged and overlapped uploads, FITS data units decoded on the GPU, tile-compressed (Rice) images, reference crop.
#pragma once
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

}  // namespace
