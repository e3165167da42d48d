// coreg_multi -- every GPU of the node from ONE process (include/coreg_hip.h, "All GPUs of the node ...").
//
// The reference's `Alignment(..., parallelism=True, counts_cpu_max=N)` fans its lag loop out over a process pool from a
// plain `python script.py` (hdrshift/alignment.py:692-744, README.md:47-87).  The same script on this library: one host
// thread + one library context (coreg_handle, own stream and buffers) per GPU, full image replicas, the lag set cut as
// euispice_coreg_amd/parallel.py cuts it for the one-process-per-GPU form (blocks of the (CRVAL1, CRVAL2) plane;
// contiguous slices of the raveled index when the plane is smaller than the number of GPUs; shares of the GRID when
// there are few lag-points per GPU), per-device sweeps with device outputs, and ONE collective -- an RCCL all-gather of
// the per-lag coefficients (all-reduce of the six sums per lag in the grid-share mode) issued for all devices between
// ncclGroupStart / ncclGroupEnd on the handles' own streams.  RCCL is looked up at run time (dlopen, an already loaded
// copy first: PyTorch ships its own); without it -- or with COREG_VIRTUAL_DEVICES, which maps several logical devices
// onto the GPUs present so that this file can be exercised on a one-GPU box -- the blocks go to the host by one
// asynchronous copy per device instead (RCCL refuses two ranks on one device).
//
// This file is included by coreg_hip.hip (same translation unit: it uses the handle's stream and staging internals).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen()ed, never linked

#include <chrono>
#include <functional>
#include <memory>
#include <queue>

namespace {

// ---- RCCL, looked up at run time ---------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: gives up a communicator whose collective never returns
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string where;
    bool ok() const { return lib != nullptr; }
    static RcclApi& get() {
        static RcclApi api = [] {
            RcclApi a;
            // COREG_RCCL_LIB: the copy that matches the HIP runtime in use (euispice_coreg_amd/_lib.py names PyTorch's
            // when it has pre-loaded PyTorch's HIP runtime); then whatever the loader finds; then the ROCm install
            const char* env = std::getenv("COREG_RCCL_LIB");
            const char* names[] = {env && *env ? env : "librccl.so", "librccl.so", "librccl.so.1",
                                   "/opt/rocm/lib/librccl.so.1"};
            // a copy some other library of this process has already loaded (PyTorch's) is preferred: two RCCL runtimes
            // in one process would each set up their own transports
            for (const char* n : names) {
                a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
                if (a.lib) {
                    a.where = std::string(n) + " (already loaded)";
                    break;
                }
            }
            if (!a.lib && !std::getenv("COREG_NO_RCCL")) {
                for (const char* n : names) {
                    a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                    if (a.lib) {
                        a.where = n;
                        break;
                    }
                }
            }
            if (std::getenv("COREG_NO_RCCL")) a.lib = nullptr;
            if (a.lib) {
#define COREG_RCCL_SYM(F) a.F = (decltype(a.F))dlsym(a.lib, "nccl" #F)
                COREG_RCCL_SYM(CommInitAll);
                COREG_RCCL_SYM(CommDestroy);
                COREG_RCCL_SYM(CommAbort);
                COREG_RCCL_SYM(AllGather);
                COREG_RCCL_SYM(AllReduce);
                COREG_RCCL_SYM(GroupStart);
                COREG_RCCL_SYM(GroupEnd);
                COREG_RCCL_SYM(GetErrorString);
#undef COREG_RCCL_SYM
                if (!a.CommInitAll || !a.CommDestroy || !a.AllGather || !a.AllReduce || !a.GroupStart || !a.GroupEnd)
                    a.lib = nullptr;
            }
            return a;
        }();
        return api;
    }
};

// ---- one host thread per device -----------------------------------------------------------------------------------
class DeviceWorker {
public:
    DeviceWorker() : t_([this] { run(); }) {}
    ~DeviceWorker() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        t_.join();
    }
    void post(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push(std::move(f));
            ++pending_;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;  // (stop_ and drained)
                f = std::move(q_.front());
                q_.pop();
            }
            f();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::queue<std::function<void()>> q_;
    int pending_ = 0;
    bool stop_ = false;
    std::thread t_;  // last: starts after the members above exist
};

// ---- the partition (mirror of euispice_coreg_amd/parallel.py: block_grid / block_bounds / lag_sharding) -----------
enum { MULTI_NONE = 0, MULTI_BLOCKS = 1, MULTI_SLICES = 2, MULTI_POINTS = 3, MULTI_COMBOS = 4 };
constexpr long long kPointShardMaxLagsPerRank = 128;  // parallel.POINT_SHARD_MAX_LAGS_PER_RANK

void multi_block_grid(int n1, int n2, int world, int* g1_out, int* g2_out) {
    int best1 = world, best2 = 1;
    long long bc0 = -1, bc1 = -1;
    for (int g1 = 1; g1 <= world; ++g1) {
        if (world % g1) continue;
        const int g2 = world / g1;
        if (g1 > n1 || g2 > n2) continue;
        const long long b1 = (n1 + g1 - 1) / g1, b2 = (n2 + g2 - 1) / g2;
        const long long c0 = std::llabs(b1 - b2), c1 = b1 * b2;
        if (bc0 < 0 || c0 < bc0 || (c0 == bc0 && c1 < bc1)) {
            bc0 = c0;
            bc1 = c1;
            best1 = g1;
            best2 = g2;
        }
    }
    *g1_out = best1;
    *g2_out = best2;
}
void multi_block_bounds(int n1, int n2, int world, int rank, int b[4]) {
    int g1, g2;
    multi_block_grid(n1, n2, world, &g1, &g2);
    const int b1 = (n1 + g1 - 1) / g1, b2 = (n2 + g2 - 1) / g2;
    const int r1 = rank / g2, r2 = rank % g2;
    b[0] = std::min(r1 * b1, n1);
    b[1] = std::min((r1 + 1) * b1, n1);
    b[2] = std::min(r2 * b2, n2);
    b[3] = std::min((r2 + 1) * b2, n2);
}
// mirror of parallel.lag_batches / combo_bounds / lag_plan (checked against them on the CPU, tests/test_host_abi.py)
constexpr double kLaunchOverheadBatches = 0.75;  // parallel.LAUNCH_OVERHEAD_BATCHES
long long multi_lag_batches(int b1, int b2) {
    if (b1 < 1 || b2 < 1) return 0;
    long long best = -1;
    for (int sw = 1; sw <= std::min(b1, 256); ++sw) {
        const int sh = std::min(b2, 256 / sw);
        const long long n = (long long)((b1 + sw - 1) / sw) * ((b2 + sh - 1) / sh);
        if (best < 0 || n < best) best = n;
    }
    return best;
}
void multi_combo_bounds(long long inner, int g_combo, int k, long long* lo, long long* hi) {
    const long long q = inner / g_combo, r = inner % g_combo;
    *lo = k * q + std::min<long long>(k, r);
    *hi = *lo + q + (k < r ? 1 : 0);
}
struct MultiPlan {
    int mode = MULTI_NONE, g_combo = 1, g1 = 1, g2 = 1;
};
MultiPlan multi_lag_plan(int n1, int n2, long long inner, int world, bool per_combo_launch = true) {
    MultiPlan p;
    const long long n = (long long)n1 * n2 * inner;
    if (world <= 1) return p;
    if (n < kPointShardMaxLagsPerRank * world) {
        p.mode = MULTI_POINTS;
        return p;
    }
    double best = -1.0;
    for (int gc = 1; gc <= world; ++gc) {
        if (world % gc || gc > inner) continue;
        const int gb = world / gc;
        bool full = true;
        for (int r = 0; r < gb && full; ++r) {
            int b[4];
            multi_block_bounds(n1, n2, gb, r, b);
            full = b[1] > b[0] && b[3] > b[2];
        }
        if (!full) continue;
        int g1, g2;
        multi_block_grid(n1, n2, gb, &g1, &g2);
        const double n_c = (double)((inner + gc - 1) / gc);
        const double n_b = (double)multi_lag_batches((n1 + g1 - 1) / g1, (n2 + g2 - 1) / g2);
        // (one launch per combination, or -- helioprojective -- one launch in all: parallel.lag_plan)
        const double cost = per_combo_launch ? n_c * (kLaunchOverheadBatches + n_b) : kLaunchOverheadBatches + n_c * n_b;
        if (best < 0 || cost < best - 1e-9 || (std::fabs(cost - best) <= 1e-9 && gc > p.g_combo)) {
            best = cost;
            p.g_combo = gc;
            p.g1 = g1;
            p.g2 = g2;
        }
    }
    if (best < 0) {
        p.mode = MULTI_SLICES;
        return p;
    }
    p.mode = p.g_combo > 1 ? MULTI_COMBOS : MULTI_BLOCKS;
    return p;
}
// device k's share under a blocks / combos plan: block b[4] of the plane, combinations [c_lo, c_hi)
void multi_grid_share(const MultiPlan& p, int n1, int n2, long long inner, int k, int b[4], long long* c_lo, long long* c_hi) {
    const int gb = p.g1 * p.g2;
    multi_block_bounds(n1, n2, gb, k % gb, b);
    multi_combo_bounds(inner, p.g_combo, k / gb, c_lo, c_hi);
}

}  // namespace

struct coreg_multi {
    int n = 0;
    bool virtual_devices = false;  // several logical devices on one GPU (COREG_VIRTUAL_DEVICES): no RCCL
    std::vector<int> devices;      // physical device of each logical one
    std::vector<coreg_handle*> h;
    std::vector<std::unique_ptr<DeviceWorker>> w;
    std::vector<ncclComm_t> comms;
    bool use_rccl = false;
    std::string collective;  // what the last sweep used, for the record
    std::string err;
    PinBuf stage;        // shared page-locked staging of an image: every device copies from it over its own link
    PinBuf host_gather;  // peer-copy collective: the devices' blocks, chunk doubles each
    std::vector<DevBuf> blk, gat;
    std::vector<DevBuf> img;  // per device: the image to align as uploaded (row shares assembled by one all-gather)
    int last_mode = MULTI_NONE;
    int force_mode = -1;            // coreg_multi_set_option "force_mode": tests of one partition on any lag set
    int opt_image_shares = 1;       // "image_shares": 1 = row shares + one all-gather when RCCL is in use, 0 = N copies
    std::string rccl_error;         // why RCCL was given up on this handle ("" = it was not)
    std::vector<hipEvent_t> pre;    // per device: recorded on the stream right before a group's collective (what the
                                    // bounded wait times is the collective, not the sweep queued in front of it)
    double phase_s[3] = {0, 0, 0};  // the last abandoned group: polling the streams, ncclCommAbort, draining the streams
    bool force_collective = false;  // COREG_MULTI_FORCE_RCCL=1 with ONE device: the RCCL calls run with a one-rank group
};

namespace {

int mfail(coreg_multi* m, int code, const std::string& msg) {
    if (m) m->err = msg;
    return code;
}

// run fn(k) on every device's own thread; first failure wins (its handle's message is kept)
int multi_run(coreg_multi* m, const std::function<int(int)>& fn) {
    std::vector<int> rc(m->n, COREG_OK);
    for (int k = 0; k < m->n; ++k) m->w[k]->post([&, k] { rc[k] = fn(k); });
    for (int k = 0; k < m->n; ++k) m->w[k]->wait();
    for (int k = 0; k < m->n; ++k)
        if (rc[k] != COREG_OK)
            return mfail(m, rc[k], "device " + std::to_string(m->devices[k]) + " (logical " + std::to_string(k) +
                                       "): " + coreg_last_error(m->h[k]));
    return COREG_OK;
}

// host image -> the shared pinned staging (one parallel copy), page-locked so that every device DMAs from it
int multi_stage(coreg_multi* m, const void* src, size_t bytes) {
    if (bytes > m->stage.cap) {
        m->stage.release();
        void* p = nullptr;
        const size_t want = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess)
            return mfail(m, COREG_ENOMEM, "hipHostMalloc (shared staging) failed");
        m->stage.p = p;
        m->stage.cap = want;
    }
    parallel_memcpy(m->stage.p, src, bytes);
    return COREG_OK;
}

int multi_sync_pivots(coreg_multi* m) {
    double piv[2];
    if (coreg_get_pivots(m->h[0], piv) != COREG_OK) return mfail(m, COREG_EHIP, coreg_last_error(m->h[0]));
    return multi_run(m, [&](int k) { return k == 0 ? COREG_OK : coreg_set_pivots(m->h[k], piv); });
}

// RCCL has failed (or never worked) on this multi-handle: from now on the blocks reach the host by one copy per device.
// The streams are drained first so that nothing of the failed group is still queued behind the sweeps.
void multi_drop_rccl(coreg_multi* m, const char* why) {
    m->use_rccl = false;
    m->rccl_error = why;
    const auto t0 = std::chrono::steady_clock::now();
    (void)multi_run(m, [&](int k) { return coreg_synchronize(m->h[k]); });
    m->phase_s[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// the stream position "just before the group": call on every device after its sweep is enqueued, before GroupStart
int multi_mark_pre_group(coreg_multi* m) {
    if (m->pre.size() != (size_t)m->n) m->pre.assign(m->n, nullptr);
    return multi_run(m, [&](int k) {
        coreg_handle* h = m->h[k];
        RETCHK(bind_device(h));
        if (!m->pre[k]) HIPCHK(hipEventCreateWithFlags(&m->pre[k], hipEventDisableTiming));
        HIPCHK(hipEventRecord(m->pre[k], h->stream));
        return COREG_OK;
    });
}

// ---- every wait on RCCL is bounded ------------------------------------------------------------
// A grouped collective is enqueued on the handles' own streams; what can fail to come back is the streams.  After each
// group the streams are polled (hipStreamQuery) for at most COREG_RCCL_WAIT_SECONDS (default 20, as the creation-time
// self-test); on expiry the communicators are aborted (ncclCommAbort releases the kernels a group holds on a stream),
// RCCL is dropped for this handle, and the caller hands the intact per-device results over through the host.
inline double multi_rccl_wait_limit() {
    const char* env = std::getenv("COREG_RCCL_WAIT_SECONDS");
    return env && std::atof(env) > 0 ? std::atof(env) : 20.0;
}

// Fault injection for the tests (COREG_RCCL_TEST_STALL=1): instead of the group's collective, every stream gets a kernel
// that spins on a flag in page-locked host memory -- a collective that never completes, as far as the stream can tell.
// The abort path sets the flag; the kernel also leaves by itself after ~12 s of GPU clock, so that a bug in the recovery
// cannot hold the device -- far enough above every wait limit the tests use (0.4 s) that a recovery released by this
// self-limit cannot pass for one released by the abort path.  The flag is COHERENT (fine-grained)
// page-locked memory, re-read with a system-scope atomic load on every turn: a host store reaches the spinning wave.
__global__ void k_test_stall(int* release, int legacy) {
    const long long t0 = wall_clock64();
    if (legacy) {  // (round 5's form, kept for the A/B of DESIGN 6: a plain volatile load of plain page-locked memory)
        volatile int* r = release;
        while (*r == 0 && wall_clock64() - t0 < 1200000000LL) __builtin_amdgcn_s_sleep(64);
        return;
    }
    while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 &&
           wall_clock64() - t0 < 1200000000LL)  // 100 MHz counter: 12 s
        __builtin_amdgcn_s_sleep(64);
}
inline int multi_test_stall_legacy() {
    const char* env = std::getenv("COREG_RCCL_TEST_STALL_LEGACY");
    return env && std::atoi(env) == 1 ? 1 : 0;
}
struct StallFlag {
    int* p = nullptr;
    StallFlag() {
        const unsigned flags = multi_test_stall_legacy() ? hipHostMallocPortable : (hipHostMallocPortable | hipHostMallocCoherent);
        if (hipHostMalloc((void**)&p, sizeof(int), flags) != hipSuccess) p = nullptr;
        if (p) __atomic_store_n(p, 0, __ATOMIC_SEQ_CST);
    }
    ~StallFlag() {
        if (p) (void)hipHostFree(p);
    }
};
inline bool multi_test_stall() {
    const char* env = std::getenv("COREG_RCCL_TEST_STALL");
    return env && std::atoi(env) == 1;
}

// true: every stream has drained.  false: at least one did not within the limit (or reported an error) -- the
// communicators are aborted and RCCL is given up for this handle; the streams are usable again on return.
bool multi_wait_group(coreg_multi* m, const char* what, StallFlag* stall = nullptr) {
    const double limit = multi_rccl_wait_limit();
    std::vector<int> state(m->n, 0);  // 1 = drained, 2 = error, 3 = timed out
    const bool have_pre = m->pre.size() == (size_t)m->n;
    const auto t_poll = std::chrono::steady_clock::now();
    (void)multi_run(m, [&](int k) {
        coreg_handle* h = m->h[k];
        RETCHK(bind_device(h));
        // the clock of the limit starts when the work queued BEFORE the group (this device's sweep) has left the stream
        // (a long sweep is not a hung collective); that part is waited for without a limit, as every
        // stream-ordered call of the library waits for its own kernels
        bool started = !(have_pre && m->pre[k]);
        auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            if (!started) {
                const hipError_t qe = hipEventQuery(m->pre[k]);
                if (qe == hipSuccess) {
                    started = true;
                    t0 = std::chrono::steady_clock::now();
                } else if (qe != hipErrorNotReady) {
                    state[k] = 2;
                    return COREG_OK;
                }
            }
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {
                state[k] = 1;
                return COREG_OK;
            }
            if (q != hipErrorNotReady) {
                state[k] = 2;
                return COREG_OK;
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (started && dt > limit) {
                state[k] = 3;
                return COREG_OK;
            }
            if (dt < 0.005) std::this_thread::yield();  // a sweep's collective takes microseconds to milliseconds
            else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    });
    bool ok = true, timed_out = false;
    for (int k = 0; k < m->n; ++k) {
        ok = ok && state[k] == 1;
        timed_out = timed_out || state[k] == 3;
    }
    if (ok) return true;
    m->phase_s[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_poll).count();
    // (test: the injected "collective" is what the abort releases -- before ncclCommAbort, which waits for the
    // communicator's stream work)
    if (stall && stall->p) __atomic_store_n(stall->p, 1, __ATOMIC_SEQ_CST);
    const auto t_abort = std::chrono::steady_clock::now();
    RcclApi& api = RcclApi::get();
    for (ncclComm_t c : m->comms)
        if (c) {
            if (api.CommAbort) (void)api.CommAbort(c);
            else (void)api.CommDestroy(c);
        }
    m->comms.clear();
    m->phase_s[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_abort).count();
    multi_drop_rccl(m, (std::string(what) + (timed_out ? " did not complete in time" : " failed on a stream")).c_str());
    // where the time of the recovery went, for whoever reads the status (coreg_multi_rccl_status)
    char buf[160];
    std::snprintf(buf, sizeof buf, " [waited %.3f s on the streams (limit %.3g s), ncclCommAbort %.3f s, stream drain %.3f s]",
                  m->phase_s[0], limit, m->phase_s[1], m->phase_s[2]);
    m->rccl_error += buf;
    return false;
}

// one sweep on every device + the collective; `launch(k, lags_k, begin, end, out_dev)` = the per-device sweep call
int multi_sweep(coreg_multi* m, const coreg_lags* lags, double* corr_out, bool per_combo_launch,
                const std::function<int(int, const coreg_lags*, int64_t, int64_t, double*)>& launch) {
    if (!m) return COREG_EINVAL;
    if (!lags || !lags->crval1 || !lags->crval2 || !lags->cdelt1 || !lags->cdelt2 || !lags->crota || lags->n_crval1 < 1 ||
        lags->n_crval2 < 1 || lags->n_cdelt1 < 1 || lags->n_cdelt2 < 1 || lags->n_crota < 1)
        return mfail(m, COREG_EINVAL, "lags: null array or empty axis");
    const int n1 = lags->n_crval1, n2 = lags->n_crval2, world = m->n;
    const long long inner = (long long)lags->n_cdelt1 * lags->n_cdelt2 * lags->n_crota;
    const long long n_lags = (long long)n1 * n2 * inner;
    if (!corr_out && n_lags > 0) return mfail(m, COREG_EINVAL, "corr_out is null");
    MultiPlan plan = multi_lag_plan(n1, n2, inner, world, per_combo_launch);
    if (m->force_mode >= 0 && world > 1) {  // tests: "force_mode" 1 = blocks, 2 = slices, 4 = combos over the whole plane
        if (m->force_mode == MULTI_SLICES) plan.mode = MULTI_SLICES;
        if (m->force_mode == MULTI_BLOCKS || m->force_mode == MULTI_COMBOS) {
            MultiPlan f;
            f.mode = m->force_mode;
            f.g_combo = m->force_mode == MULTI_COMBOS ? (int)std::min<long long>(world, inner) : 1;
            while (world % f.g_combo) --f.g_combo;
            multi_block_grid(n1, n2, world / f.g_combo, &f.g1, &f.g2);
            bool full = true;
            for (int r = 0; r < f.g1 * f.g2 && full; ++r) {
                int bb[4];
                multi_block_bounds(n1, n2, f.g1 * f.g2, r, bb);
                full = bb[1] > bb[0] && bb[3] > bb[2];
            }
            if (full) plan = f;
        }
    }
    int mode = plan.mode;
    if (mode == MULTI_NONE && m->force_collective) mode = MULTI_SLICES;  // one slice, one-rank all-gather
    m->last_mode = mode;
    if (mode == MULTI_NONE) {
        m->collective = "none";
        int rc = COREG_OK;
        m->w[0]->post([&] { rc = launch(0, lags, 0, n_lags, nullptr); });
        m->w[0]->wait();
        if (rc != COREG_OK) return mfail(m, rc, coreg_last_error(m->h[0]));
        return COREG_OK;
    }
    bool rccl = m->use_rccl;
    m->collective = rccl ? "rccl" : "host-copy";

    if (mode == MULTI_POINTS) {
        // every device sweeps ALL lag-points over its share of the grid; the six sums per lag slot are added over the
        // devices (all-reduce), device 0 evaluates the coefficients.  Whatever happens in between, every context is
        // back to "unsharded" and drained when this branch is left (a later sweep on the same contexts must not run as
        // a grid share).
        struct Unshard {
            coreg_multi* m;
            ~Unshard() {
                (void)multi_run(m, [&](int k) {
                    (void)coreg_set_option(m->h[k], "shard_world", 1);
                    return coreg_synchronize(m->h[k]);
                });
            }
        } unshard{m};
        RETCHK(multi_sync_pivots(m));
        RETCHK(multi_run(m, [&](int k) {
            coreg_handle* h = m->h[k];
            RETCHK(bind_device(h));
            RETCHK(coreg_set_option(h, "shard_world", world));
            RETCHK(coreg_set_option(h, "shard_rank", k));
            HIPCHK(m->blk[k].reserve((size_t)std::max<long long>(n_lags, 1) * sizeof(double)));
            return launch(k, lags, 0, n_lags, m->blk[k].as<double>());
        }));
        int rc = COREG_OK;
        int64_t n6 = 0;
        coreg_sums_size(m->h[0], &n6);
        bool reduced_on_device = false;
        if (rccl && n6 > 0) {
            // OUT of place (into gat[k]): whatever happens to the collective, every device's own sums stay intact and
            // the host path below can still add them
            RETCHK(multi_run(m, [&](int k) {
                coreg_handle* h = m->h[k];
                RETCHK(bind_device(h));
                HIPCHK(m->gat[k].reserve((size_t)n6 * sizeof(double)));
                return COREG_OK;
            }));
            StallFlag stall;
            ncclResult_t e = ncclSuccess, e2 = ncclSuccess;
            RETCHK(multi_mark_pre_group(m));
            if (multi_test_stall() && stall.p) {
                (void)multi_run(m, [&](int k) {
                    RETCHK(bind_device(m->h[k]));
                    hipLaunchKernelGGL(k_test_stall, dim3(1), dim3(1), 0, m->h[k]->stream, stall.p, multi_test_stall_legacy());
                    return COREG_OK;
                });
            } else {
                e = RcclApi::get().GroupStart();
                for (int k = 0; k < world && e == ncclSuccess; ++k)
                    e = RcclApi::get().AllReduce(m->h[k]->sums.p, m->gat[k].p, (size_t)n6, ncclDouble, ncclSum, m->comms[k],
                                                 m->h[k]->stream);
                e2 = RcclApi::get().GroupEnd();
            }
            if (e != ncclSuccess || e2 != ncclSuccess) {
                multi_drop_rccl(m, "RCCL all-reduce failed");
                m->collective = "host-copy (RCCL all-reduce failed)";
            } else if (!multi_wait_group(m, "RCCL all-reduce", &stall) || multi_test_stall()) {
                m->collective = "host-copy (RCCL all-reduce did not complete)";
            } else {
                reduced_on_device = true;
            }
        }
        if (reduced_on_device) {
            m->w[0]->post([&] { rc = coreg_finalize_sums(m->h[0], m->gat[0].as<double>(), 1, corr_out, 0); });
            m->w[0]->wait();
            if (rc != COREG_OK) mfail(m, rc, coreg_last_error(m->h[0]));
        } else if (n6 > 0) {
            std::vector<std::vector<double>> part(world, std::vector<double>((size_t)n6));
            rc = multi_run(m, [&](int k) { return coreg_copy_sums(m->h[k], part[k].data(), 0); });
            if (rc == COREG_OK) {
                for (int k = 1; k < world; ++k)  // fixed order: deterministic
                    for (int64_t i = 0; i < n6; ++i) part[0][(size_t)i] += part[k][(size_t)i];
                m->w[0]->post([&] { rc = coreg_finalize_sums(m->h[0], part[0].data(), 0, corr_out, 0); });
                m->w[0]->wait();
                if (rc != COREG_OK) mfail(m, rc, coreg_last_error(m->h[0]));
            }
        } else {
            // no launch happened anywhere (every lag-point invalid): the NaN-filled output of device 0
            m->w[0]->post([&] {
                rc = bind_device(m->h[0]);
                if (rc == COREG_OK && hipMemcpyAsync(corr_out, m->blk[0].p, (size_t)n_lags * sizeof(double),
                                                     hipMemcpyDeviceToHost, m->h[0]->stream) != hipSuccess)
                    rc = COREG_EHIP;
                if (rc == COREG_OK && hipStreamSynchronize(m->h[0]->stream) != hipSuccess) rc = COREG_EHIP;
            });
            m->w[0]->wait();
        }
        return rc;  // (~Unshard: back to unsharded contexts, every stream drained before the caller's buffers go)
    }

    // ---- lag sharding: a g_combo x (g1 x g2) grid of (combination run) x (block of the CRVAL plane), or contiguous
    //      slices of the raveled index
    const bool grid = mode == MULTI_BLOCKS || mode == MULTI_COMBOS;
    long long chunk;
    if (grid)
        chunk = (long long)((n1 + plan.g1 - 1) / plan.g1) * ((n2 + plan.g2 - 1) / plan.g2) *
                ((inner + plan.g_combo - 1) / plan.g_combo);
    else
        chunk = (n_lags + world - 1) / world;
    if (m->host_gather.reserve((size_t)chunk * world * sizeof(double)) != hipSuccess)
        return mfail(m, COREG_ENOMEM, "hipHostMalloc (gather buffer) failed");
    std::vector<long long> n_mine(world, 0);
    auto to_host = [&](int k) {  // device k's block straight to its place in the host buffer, and its stream drained
        coreg_handle* h = m->h[k];
        RETCHK(bind_device(h));
        if (n_mine[k] > 0)
            HIPCHK(hipMemcpyAsync((double*)m->host_gather.p + (size_t)k * chunk, m->blk[k].p,
                                  (size_t)n_mine[k] * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return COREG_OK;
    };
    RETCHK(multi_run(m, [&](int k) {
        coreg_handle* h = m->h[k];
        RETCHK(bind_device(h));
        RETCHK(coreg_set_option(h, "shard_world", 1));  // (never a grid share here, whatever an earlier call left)
        HIPCHK(m->blk[k].reserve((size_t)chunk * sizeof(double)));
        if (rccl) HIPCHK(m->gat[k].reserve((size_t)chunk * world * sizeof(double)));
        coreg_lags sub = *lags;
        int64_t lo = 0, hi = 0;
        if (grid) {
            int b[4];
            long long c_lo, c_hi;
            multi_grid_share(plan, n1, n2, inner, k, b, &c_lo, &c_hi);
            sub.crval1 = lags->crval1 + b[0];
            sub.n_crval1 = b[1] - b[0];
            sub.crval2 = lags->crval2 + b[2];
            sub.n_crval2 = b[3] - b[2];
            hi = (int64_t)sub.n_crval1 * sub.n_crval2 * (c_hi - c_lo);
            if (hi > 0 && (c_lo != 0 || c_hi != inner)) {  // one-shot: consumed by the launch below
                RETCHK(coreg_set_option(h, "combo_begin", c_lo));
                RETCHK(coreg_set_option(h, "combo_end", c_hi));
            }
        } else {
            lo = std::min<long long>((long long)k * chunk, n_lags);
            hi = std::min<long long>((long long)(k + 1) * chunk, n_lags);
        }
        n_mine[k] = hi - lo;
        if (hi > lo) RETCHK(launch(k, &sub, lo, hi, m->blk[k].as<double>()));
        if (!rccl) RETCHK(to_host(k));
        return COREG_OK;
    }));
    if (rccl) {
        // THE collective: one all-gather of `chunk` doubles per device, all devices in one group, each on the stream its
        // sweep was enqueued on
        StallFlag stall;
        ncclResult_t e = ncclSuccess, e2 = ncclSuccess;
        RETCHK(multi_mark_pre_group(m));
        if (multi_test_stall() && stall.p) {
            (void)multi_run(m, [&](int k) {
                RETCHK(bind_device(m->h[k]));
                hipLaunchKernelGGL(k_test_stall, dim3(1), dim3(1), 0, m->h[k]->stream, stall.p, multi_test_stall_legacy());
                return COREG_OK;
            });
        } else {
            e = RcclApi::get().GroupStart();
            for (int k = 0; k < world && e == ncclSuccess; ++k)
                e = RcclApi::get().AllGather(m->blk[k].p, m->gat[k].p, (size_t)chunk, ncclDouble, m->comms[k], m->h[k]->stream);
            e2 = RcclApi::get().GroupEnd();
        }
        if (e != ncclSuccess || e2 != ncclSuccess) {
            // the blocks themselves are intact on their devices: hand them over through the host instead, now and for
            // every later sweep of this handle
            multi_drop_rccl(m, "RCCL all-gather failed");
            m->collective = "host-copy (RCCL all-gather failed)";
            rccl = false;
            RETCHK(multi_run(m, to_host));
        } else if (!multi_wait_group(m, "RCCL all-gather", &stall) || multi_test_stall()) {
            // a group that did not come back within the limit: aborted; the blocks are intact on their devices
            m->collective = "host-copy (RCCL all-gather did not complete)";
            rccl = false;
            RETCHK(multi_run(m, to_host));
        } else {
            RETCHK(multi_run(m, [&](int k) {
                coreg_handle* h = m->h[k];
                RETCHK(bind_device(h));
                if (k == 0)  // every device holds the whole map; device 0 hands it to the host
                    HIPCHK(hipMemcpyAsync(m->host_gather.p, m->gat[0].p, (size_t)chunk * world * sizeof(double),
                                          hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                return COREG_OK;
            }));
        }
    }
    const double* gathered = (const double*)m->host_gather.p;
    // gathered chunks -> C-order raveled map
    if (!grid) {
        for (int k = 0; k < world; ++k)
            if (n_mine[k] > 0)
                std::memcpy(corr_out + (size_t)k * chunk, gathered + (size_t)k * chunk, (size_t)n_mine[k] * sizeof(double));
    } else {
        for (int k = 0; k < world; ++k) {
            int b[4];
            long long c_lo, c_hi;
            multi_grid_share(plan, n1, n2, inner, k, b, &c_lo, &c_hi);
            const int w2 = b[3] - b[2];
            const long long ncm = c_hi - c_lo;
            if (ncm <= 0) continue;
            for (int i1 = b[0]; i1 < b[1]; ++i1)
                for (int i2 = b[2]; i2 < b[3]; ++i2)
                    std::memcpy(corr_out + ((size_t)i1 * n2 + i2) * inner + c_lo,
                                gathered + (size_t)k * chunk + ((size_t)(i1 - b[0]) * w2 + (i2 - b[2])) * ncm,
                                (size_t)ncm * sizeof(double));
        }
    }
    return COREG_OK;
}

// The image to align crosses PCIe ONCE in all: device k uploads rows [k * ceil(H / N), ...) over its own link, ONE
// all-gather over xGMI (in place: every device's share sits where the gathered image wants it) assembles the replica on
// every GPU, and each device adopts (and, for raw FITS bytes, decodes) its copy.  The reference hands its workers the
// image through ONE shared-memory copy (alignment.py:657-665).  Returns COREG_ENOTIMPL when RCCL is not in use or
// fails (the caller then stages the whole image once in page-locked memory and every device copies it).
int multi_set_small_shares(coreg_multi* m, const void* src, const PixFmt& fmt, const coreg_fits_pixels* px, int dtype,
                           int32_t ny, int32_t nx) {
    const int world = m->n;
    if (!m->use_rccl || m->comms.size() != (size_t)world) return COREG_ENOTIMPL;
    const size_t row_bytes = (size_t)nx * fmt.elem();
    const size_t rows = ((size_t)ny + world - 1) / world, share = rows * row_bytes;
    if (share == 0 || share > ((size_t)1 << 40)) return COREG_ENOTIMPL;
    RETCHK(multi_run(m, [&](int k) {
        coreg_handle* h = m->h[k];
        RETCHK(bind_device(h));
        HIPCHK(m->img[k].reserve(share * world));
        const size_t lo = std::min<size_t>((size_t)k * rows, (size_t)ny), hi = std::min<size_t>(lo + rows, (size_t)ny);
        if (hi > lo)
            RETCHK(staged_upload(h, (char*)m->img[k].p + (size_t)k * share, (const char*)src + lo * row_bytes,
                                 (hi - lo) * row_bytes));
        return COREG_OK;
    }));
    RETCHK(multi_mark_pre_group(m));
    RcclApi& api = RcclApi::get();
    ncclResult_t e = api.GroupStart();
    for (int k = 0; k < world && e == ncclSuccess; ++k)
        e = api.AllGather((const char*)m->img[k].p + (size_t)k * share, m->img[k].p, share, ncclChar, m->comms[k],
                          m->h[k]->stream);
    const ncclResult_t e2 = api.GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess) {
        multi_drop_rccl(m, "RCCL all-gather of the image failed");
        return COREG_ENOTIMPL;
    }
    if (!multi_wait_group(m, "RCCL all-gather of the image")) return COREG_ENOTIMPL;  // (the caller stages the whole image)
    return multi_run(m, [&](int k) {
        coreg_handle* h = m->h[k];
        if (px) {
            coreg_fits_pixels dev = *px;
            dev.data = m->img[k].p;
            RETCHK(set_small_fits(h, &dev, ny, nx, SRC_DEVICE));
        } else {
            RETCHK(set_small_direct(h, m->img[k].p, dtype, ny, nx, SRC_DEVICE));
        }
        return coreg_synchronize(h);  // the caller's host buffer and the share buffers are free again on return
    });
}

// The communicators have never carried data: before any result depends on them, every device contributes (k + 1) * 1.5
// to ONE grouped all-gather -- the very call a sweep makes -- and checks what it received.  The wait is bounded
// (COREG_RCCL_SELFTEST_SECONDS, default 20): a group that does not come back is aborted (ncclCommAbort) and the handle
// falls back to host copies, as it does when the pattern is wrong or a call fails.
void multi_rccl_selftest(coreg_multi* m) {
    const int world = m->n;
    RcclApi& api = RcclApi::get();
    std::string why;
    bool ok = multi_run(m, [&](int k) {
                  coreg_handle* h = m->h[k];
                  RETCHK(bind_device(h));
                  HIPCHK(m->blk[k].reserve(sizeof(double)));
                  HIPCHK(m->gat[k].reserve((size_t)world * sizeof(double)));
                  const double v = (k + 1) * 1.5;
                  HIPCHK(hipMemcpyAsync(m->blk[k].p, &v, sizeof(double), hipMemcpyHostToDevice, h->stream));
                  HIPCHK(hipMemsetAsync(m->gat[k].p, 0, (size_t)world * sizeof(double), h->stream));
                  HIPCHK(hipStreamSynchronize(h->stream));
                  return COREG_OK;
              }) == COREG_OK;
    if (!ok) why = "self-test setup failed";
    if (ok) {
        ncclResult_t e = api.GroupStart();
        for (int k = 0; k < world && e == ncclSuccess; ++k)
            e = api.AllGather(m->blk[k].p, m->gat[k].p, 1, ncclDouble, m->comms[k], m->h[k]->stream);
        const ncclResult_t e2 = api.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess) {
            ok = false;
            why = "self-test all-gather returned an error";
        }
    }
    bool timed_out = false;
    if (ok) {
        const char* env = std::getenv("COREG_RCCL_SELFTEST_SECONDS");
        const double limit = env && std::atof(env) > 0 ? std::atof(env) : 20.0;
        std::vector<int> state(world, 0);  // 1 = done and right, 2 = done and wrong, 3 = timed out
        (void)multi_run(m, [&](int k) {
            coreg_handle* h = m->h[k];
            RETCHK(bind_device(h));
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                const hipError_t q = hipStreamQuery(h->stream);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) {
                    state[k] = 2;
                    return COREG_OK;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                    state[k] = 3;
                    return COREG_OK;
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            std::vector<double> got((size_t)world);
            if (hipMemcpy(got.data(), m->gat[k].p, (size_t)world * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
                state[k] = 2;
                return COREG_OK;
            }
            state[k] = 1;
            for (int j = 0; j < world; ++j)
                if (got[(size_t)j] != (j + 1) * 1.5) state[k] = 2;
            return COREG_OK;
        });
        for (int k = 0; k < world; ++k) {
            if (state[k] == 3) timed_out = true;
            if (state[k] != 1) ok = false;
        }
        if (!ok) why = timed_out ? "self-test all-gather did not complete in time" : "self-test all-gather gave wrong values";
    }
    if (ok) return;
    // give the communicators up; an aborted group releases the streams it holds
    for (ncclComm_t c : m->comms)
        if (c) {
            if (timed_out && api.CommAbort) (void)api.CommAbort(c);
            else (void)api.CommDestroy(c);
        }
    m->comms.clear();
    m->use_rccl = false;
    m->rccl_error = why;
}

}  // namespace

extern "C" {

int coreg_device_count(void) {
    const char* v = std::getenv("COREG_VIRTUAL_DEVICES");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (v && std::atoi(v) > 0 && ndev > 0) return std::atoi(v);
    return ndev;
}

int coreg_physical_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    return ndev;
}

int coreg_multi_plan(int32_t n_crval1, int32_t n_crval2, int64_t n_inner, int32_t world, int32_t per_combo_launch,
                     int32_t* mode, int32_t* g_combo, int32_t* g1, int32_t* g2) {
    if (n_crval1 < 1 || n_crval2 < 1 || n_inner < 1 || world < 1 || !mode || !g_combo || !g1 || !g2) return COREG_EINVAL;
    const MultiPlan p = multi_lag_plan(n_crval1, n_crval2, n_inner, world, per_combo_launch != 0);
    *mode = p.mode;
    *g_combo = p.g_combo;
    *g1 = p.g1;
    *g2 = p.g2;
    if (p.mode != MULTI_BLOCKS && p.mode != MULTI_COMBOS) {  // (what a block partition would be, for the record)
        int a, b;
        multi_block_grid(n_crval1, n_crval2, world, &a, &b);
        *g1 = a;
        *g2 = b;
    }
    return COREG_OK;
}

int coreg_multi_create(coreg_multi** out, int n_devices, const int* device_ids) {
    if (!out || n_devices < 0 || n_devices > kMaxDevices) return COREG_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return COREG_EHIP;
    const char* v = std::getenv("COREG_VIRTUAL_DEVICES");
    const int n_virtual = (v && std::atoi(v) > 0) ? std::min(std::atoi(v), kMaxDevices) : 0;
    coreg_multi* m = new (std::nothrow) coreg_multi();
    if (!m) return COREG_ENOMEM;
    m->virtual_devices = n_virtual > 0 && !device_ids;
    m->n = n_devices > 0 ? n_devices : (m->virtual_devices ? n_virtual : ndev);
    for (int k = 0; k < m->n; ++k) {
        const int d = device_ids ? device_ids[k] : (m->virtual_devices ? k % ndev : k);
        if (d < 0 || d >= ndev) {
            delete m;
            return COREG_EINVAL;
        }
        m->devices.push_back(d);
    }
    bool distinct = true;
    for (int a = 0; a < m->n; ++a)
        for (int b = a + 1; b < m->n; ++b) distinct = distinct && m->devices[a] != m->devices[b];
    m->h.assign(m->n, nullptr);
    m->blk.resize(m->n);
    m->gat.resize(m->n);
    m->img.resize(m->n);
    for (int k = 0; k < m->n; ++k) m->w.emplace_back(new DeviceWorker());
    std::vector<int> rc(m->n, COREG_OK);
    for (int k = 0; k < m->n; ++k) m->w[k]->post([&, k] { rc[k] = coreg_create(&m->h[k], m->devices[k]); });
    for (int k = 0; k < m->n; ++k) m->w[k]->wait();
    for (int k = 0; k < m->n; ++k)
        if (rc[k] != COREG_OK) {
            const int r = rc[k];
            coreg_multi_destroy(m);
            return r;
        }
    // one communicator per device, created together (ncclCommInitAll: the single-process form).  COREG_MULTI_COLLECTIVE=host
    // keeps RCCL out of it (the blocks then reach the host by one copy per device, a few KB each).
    const char* force = std::getenv("COREG_MULTI_FORCE_RCCL");
    const char* coll = std::getenv("COREG_MULTI_COLLECTIVE");
    const bool want_rccl = !(coll && std::string(coll) == "host");
    m->force_collective = m->n == 1 && force && std::atoi(force) == 1;
    if (want_rccl && (m->n > 1 || m->force_collective) && distinct && RcclApi::get().ok()) {
        // ncclCommInitAll on a helper thread, waited for with the self-test's limit: a bootstrap that never returns costs
        // this handle RCCL (host copies instead), never the process.  The thread owns its arguments (shared state), so
        // that it may outlive this function; it is detached only when it has not come back in time.
        struct InitState {
            std::vector<ncclComm_t> comms;
            std::vector<int> devs;
            std::mutex mu;
            std::condition_variable cv;
            bool done = false;
            bool abandoned = false;  // the creator gave up waiting: whatever the thread still obtains is its own to free
            ncclResult_t rc = ncclSuccess;
        };
        auto st = std::make_shared<InitState>();
        st->comms.assign(m->n, nullptr);
        st->devs = m->devices;
        std::thread init([st] {
            const char* stall = std::getenv("COREG_RCCL_TEST_INIT_STALL_SECONDS");  // tests: a bootstrap that hangs
            if (stall && std::atof(stall) > 0)
                std::this_thread::sleep_for(std::chrono::milliseconds((long long)(std::atof(stall) * 1000)));
            const ncclResult_t r = RcclApi::get().CommInitAll(st->comms.data(), (int)st->devs.size(), st->devs.data());
            bool abandoned;
            {
                std::lock_guard<std::mutex> lk(st->mu);
                st->rc = r;
                st->done = true;
                abandoned = st->abandoned;
                st->cv.notify_all();
            }
            if (abandoned && r == ncclSuccess) {
                // nobody will ever use these communicators -- the late thread gives them back itself
                // (abort: nothing is queued on them) instead of leaving them to process teardown
                RcclApi& api = RcclApi::get();
                for (ncclComm_t c : st->comms)
                    if (c) {
                        if (api.CommAbort) (void)api.CommAbort(c);
                        else (void)api.CommDestroy(c);
                    }
                st->comms.clear();
            }
        });
        const char* env = std::getenv("COREG_RCCL_SELFTEST_SECONDS");
        const double limit = env && std::atof(env) > 0 ? std::atof(env) : 20.0;
        bool in_time;
        {
            std::unique_lock<std::mutex> lk(st->mu);
            in_time = st->cv.wait_for(lk, std::chrono::milliseconds((long long)(limit * 1000)), [&] { return st->done; });
            if (!in_time) st->abandoned = true;  // (under the lock: the thread either sees it or has already finished)
        }
        if (!in_time) {
            // its communicators, if it ever gets them, are never used: it frees them itself.  Until it has come back the
            // process should not count on a clean exit from inside RCCL's bootstrap (documented in include/coreg_hip.h)
            init.detach();
            m->rccl_error = "ncclCommInitAll did not return in time";
        } else {
            init.join();
            if (st->rc == ncclSuccess) {
                m->comms = st->comms;
                m->use_rccl = true;
                multi_rccl_selftest(m);  // a group that does not gather a known pattern is not used for results
            } else {
                m->rccl_error = "ncclCommInitAll failed";
            }
        }
    }
    m->collective = m->use_rccl ? "rccl" : (m->n > 1 ? "host-copy" : "none");
    *out = m;
    return COREG_OK;
}

void coreg_multi_destroy(coreg_multi* m) {
    if (!m) return;
    for (int k = 0; k < m->n; ++k) {
        if (!m->h[k]) continue;
        m->w[k]->post([m, k] {
            (void)hipSetDevice(m->devices[k]);
            if (m->h[k]->stream) (void)hipStreamSynchronize(m->h[k]->stream);
            m->blk[k].release();
            m->gat[k].release();
            m->img[k].release();
            if (k < (int)m->pre.size() && m->pre[k]) (void)hipEventDestroy(m->pre[k]);
        });
        m->w[k]->wait();
    }
    for (ncclComm_t c : m->comms)  // (empty when RCCL was never set up or has been given up)
        if (c) (void)RcclApi::get().CommDestroy(c);
    m->comms.clear();
    for (int k = 0; k < m->n; ++k) {
        if (!m->h[k]) continue;
        m->w[k]->post([m, k] { coreg_destroy(m->h[k]); });
        m->w[k]->wait();
    }
    m->stage.release();
    m->host_gather.release();
    m->w.clear();  // joins the threads
    delete m;
}

int coreg_multi_size(const coreg_multi* m) { return m ? m->n : 0; }
coreg_handle* coreg_multi_handle(coreg_multi* m, int k) { return (m && k >= 0 && k < m->n) ? m->h[k] : nullptr; }
const char* coreg_multi_last_error(const coreg_multi* m) { return m ? m->err.c_str() : "null multi-handle"; }
const char* coreg_multi_collective(const coreg_multi* m) { return m ? m->collective.c_str() : ""; }
const char* coreg_multi_rccl_status(const coreg_multi* m) {
    if (!m) return "";
    if (m->use_rccl) return "ok";
    return m->rccl_error.empty() ? "not used" : m->rccl_error.c_str();
}
int coreg_multi_last_mode(const coreg_multi* m) { return m ? m->last_mode : 0; }

int coreg_multi_set_option(coreg_multi* m, const char* name, int64_t value) {
    if (!m || !name) return COREG_EINVAL;
    if (std::string(name) == "force_mode") {  // -1 = the planner decides
        if (value != -1 && value != MULTI_BLOCKS && value != MULTI_SLICES && value != MULTI_COMBOS)
            return mfail(m, COREG_EINVAL, "force_mode must be -1, 1 (blocks), 2 (slices) or 4 (combos)");
        m->force_mode = (int)value;
        return COREG_OK;
    }
    if (std::string(name) == "image_shares") {
        m->opt_image_shares = value ? 1 : 0;
        return COREG_OK;
    }
    return multi_run(m, [&](int k) { return coreg_set_option(m->h[k], name, value); });
}

int coreg_multi_set_small(coreg_multi* m, const void* img, int dtype, int32_t ny, int32_t nx) {
    if (!m) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || (dtype != COREG_F32 && dtype != COREG_F64))
        return mfail(m, COREG_EINVAL, "multi_set_small: bad argument");
    if (m->opt_image_shares) {
        const int rc = multi_set_small_shares(m, img, PixFmt::native(dtype == COREG_F32), nullptr, dtype, ny, nx);
        if (rc != COREG_ENOTIMPL) return rc;
    }
    if (m->n == 1) {
        int rc;
        m->w[0]->post([&] {
            rc = dtype == COREG_F32 ? coreg_set_small_f32(m->h[0], (const float*)img, ny, nx)
                                    : coreg_set_small(m->h[0], (const double*)img, ny, nx);
        });
        m->w[0]->wait();
        return rc == COREG_OK ? rc : mfail(m, rc, coreg_last_error(m->h[0]));
    }
    RETCHK(multi_stage(m, img, (size_t)ny * nx * (dtype == COREG_F32 ? 4 : 8)));
    return multi_run(m, [&](int k) {
        RETCHK(set_small_direct(m->h[k], m->stage.p, dtype, ny, nx, SRC_PINNED));
        return coreg_synchronize(m->h[k]);  // the shared staging is free again on return
    });
}

int coreg_multi_threshold_small(coreg_multi* m, int has_min, double vmin, int has_max, double vmax, long long* n_finite) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) {
        return coreg_threshold_small(m->h[k], has_min, vmin, has_max, vmax, k == 0 ? n_finite : nullptr);
    });
}

int coreg_multi_set_reference_on_grid(coreg_multi* m, const void* ref, int dtype, int32_t gy, int32_t gx) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) { return coreg_set_reference_on_grid(m->h[k], ref, dtype, gy, gx); });
}

int coreg_multi_prepare_reference_carrington(coreg_multi* m, const void* large, int dtype, int32_t ny, int32_t nx,
                                             const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid, double solar_r,
                                             int order) {
    if (!m) return COREG_EINVAL;
    if (!large || ny < 1 || nx < 1 || (dtype != COREG_F32 && dtype != COREG_F64))
        return mfail(m, COREG_EINVAL, "multi_prepare_reference: bad argument");
    // every device sends its own copy of the rectangle the grid can touch (reference_crop: usually a few hundred KB;
    // when the grid covers most of the image, N whole uploads through the devices' own staging buffers)
    return multi_run(m, [&](int k) {
        return prepare_carrington(m->h[k], large, PixFmt::native(dtype == COREG_F32), ny, nx, hdr_large, grid, solar_r,
                                  order, SRC_HOST);
    });
}

int coreg_multi_prepare_reference_helioprojective(coreg_multi* m, const void* large, int dtype, int32_t ny, int32_t nx,
                                                  const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small, int order) {
    if (!m) return COREG_EINVAL;
    if (!large || ny < 1 || nx < 1 || (dtype != COREG_F32 && dtype != COREG_F64))
        return mfail(m, COREG_EINVAL, "multi_prepare_reference: bad argument");
    return multi_run(m, [&](int k) {
        return prepare_helioprojective(m->h[k], large, PixFmt::native(dtype == COREG_F32), ny, nx, hdr_large, hdr_small,
                                       order, SRC_HOST);
    });
}

// the same three with the pixels as the FITS data unit stores them (raw big-endian bytes up, decode on each GPU)
int coreg_multi_set_small_fits(coreg_multi* m, const coreg_fits_pixels* px, int32_t ny, int32_t nx) {
    if (!m) return COREG_EINVAL;
    PixFmt fmt;
    if (check_fits(nullptr, px, &fmt) != COREG_OK || ny < 1 || nx < 1)
        return mfail(m, COREG_EINVAL, "multi_set_small_fits: bad argument");
    if (m->opt_image_shares) {
        const int rc = multi_set_small_shares(m, px->data, fmt, px, 0, ny, nx);
        if (rc != COREG_ENOTIMPL) return rc;
    }
    if (m->n == 1) {
        int rc;
        m->w[0]->post([&] { rc = coreg_set_small_fits(m->h[0], px, ny, nx); });
        m->w[0]->wait();
        return rc == COREG_OK ? rc : mfail(m, rc, coreg_last_error(m->h[0]));
    }
    RETCHK(multi_stage(m, px->data, (size_t)ny * nx * fmt.elem()));
    coreg_fits_pixels staged = *px;
    staged.data = m->stage.p;
    return multi_run(m, [&](int k) {
        RETCHK(set_small_fits(m->h[k], &staged, ny, nx, SRC_PINNED));
        return coreg_synchronize(m->h[k]);  // the shared staging is free again on return
    });
}

int coreg_multi_prepare_reference_carrington_fits(coreg_multi* m, const coreg_fits_pixels* px, int32_t ny, int32_t nx,
                                                  const coreg_wcs2d* hdr_large, const coreg_carr_grid* grid,
                                                  double solar_r, int order) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) {
        return coreg_prepare_reference_carrington_fits(m->h[k], px, ny, nx, hdr_large, grid, solar_r, order);
    });
}

int coreg_multi_prepare_reference_helioprojective_fits(coreg_multi* m, const coreg_fits_pixels* px, int32_t ny,
                                                       int32_t nx, const coreg_wcs2d* hdr_large,
                                                       const coreg_wcs2d* hdr_small, int order) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) {
        return coreg_prepare_reference_helioprojective_fits(m->h[k], px, ny, nx, hdr_large, hdr_small, order);
    });
}

// ... and as a tile-compressed image (every device uploads the compressed bytes itself -- a quarter of the pixels -- and
// decodes them)
int coreg_multi_set_small_tiled(coreg_multi* m, const coreg_fits_tiled* t) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) { return coreg_set_small_tiled(m->h[k], t); });
}

int coreg_multi_prepare_reference_carrington_tiled(coreg_multi* m, const coreg_fits_tiled* t, const coreg_wcs2d* hdr_large,
                                                   const coreg_carr_grid* grid, double solar_r, int order) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) {
        return coreg_prepare_reference_carrington_tiled(m->h[k], t, hdr_large, grid, solar_r, order);
    });
}

int coreg_multi_prepare_reference_helioprojective_tiled(coreg_multi* m, const coreg_fits_tiled* t,
                                                        const coreg_wcs2d* hdr_large, const coreg_wcs2d* hdr_small,
                                                        int order) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) {
        return coreg_prepare_reference_helioprojective_tiled(m->h[k], t, hdr_large, hdr_small, order);
    });
}

int coreg_multi_sweep_carrington(coreg_multi* m, const coreg_wcs2d* hdr_small, const coreg_carr_grid* grid, double solar_r,
                                 const coreg_lags* lags, int order, int method, int cdelt_semantics, double* corr_out) {
    return multi_sweep(m, lags, corr_out, true, [&](int k, const coreg_lags* l, int64_t lo, int64_t hi, double* out_dev) {
        return out_dev ? coreg_sweep_carrington(m->h[k], hdr_small, grid, solar_r, l, order, method, cdelt_semantics, lo,
                                                hi, out_dev, 1)
                       : coreg_sweep_carrington(m->h[k], hdr_small, grid, solar_r, l, order, method, cdelt_semantics, lo,
                                                hi, corr_out, 0);
    });
}

int coreg_multi_sweep_helioprojective(coreg_multi* m, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small,
                                      const coreg_lags* lags, int order, int method, int cdelt_semantics, double* corr_out) {
    // (TAN headers: one launch whatever the lag set; plate-carree maps: one launch per (cdelt, crota) combination)
    const bool per_combo = hdr_target && hdr_target->proj == COREG_PROJ_CAR;
    return multi_sweep(m, lags, corr_out, per_combo, [&](int k, const coreg_lags* l, int64_t lo, int64_t hi, double* out_dev) {
        return out_dev ? coreg_sweep_helioprojective(m->h[k], hdr_target, hdr_small, l, order, method, cdelt_semantics, lo,
                                                     hi, out_dev, 1)
                       : coreg_sweep_helioprojective(m->h[k], hdr_target, hdr_small, l, order, method, cdelt_semantics, lo,
                                                     hi, corr_out, 0);
    });
}

int coreg_multi_last_stats(coreg_multi* m, int k, coreg_stats* out) {
    if (!m || k < 0 || k >= m->n || !out) return COREG_EINVAL;
    int rc;
    m->w[k]->post([&] { rc = coreg_last_stats(m->h[k], out); });
    m->w[k]->wait();
    return rc;
}

}  // extern "C"
