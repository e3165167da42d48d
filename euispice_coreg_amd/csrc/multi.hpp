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

#include <functional>
#include <memory>
#include <queue>

namespace {

// ---- RCCL, looked up at run time ---------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
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
enum { MULTI_NONE = 0, MULTI_BLOCKS = 1, MULTI_SLICES = 2, MULTI_POINTS = 3 };
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
int multi_lag_sharding(int n1, int n2, long long inner, int world) {
    const long long n = (long long)n1 * n2 * inner;
    if (world <= 1) return MULTI_NONE;
    if (n < kPointShardMaxLagsPerRank * world) return MULTI_POINTS;
    for (int r = 0; r < world; ++r) {
        int b[4];
        multi_block_bounds(n1, n2, world, r, b);
        if (!(b[1] > b[0] && b[3] > b[2])) return MULTI_SLICES;
    }
    return MULTI_BLOCKS;
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
    int last_mode = MULTI_NONE;
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

// one sweep on every device + the collective; `launch(k, lags_k, begin, end, out_dev)` = the per-device sweep call
int multi_sweep(coreg_multi* m, const coreg_lags* lags, double* corr_out,
                const std::function<int(int, const coreg_lags*, int64_t, int64_t, double*)>& launch) {
    if (!m) return COREG_EINVAL;
    if (!lags || !lags->crval1 || !lags->crval2 || !lags->cdelt1 || !lags->cdelt2 || !lags->crota || lags->n_crval1 < 1 ||
        lags->n_crval2 < 1 || lags->n_cdelt1 < 1 || lags->n_cdelt2 < 1 || lags->n_crota < 1)
        return mfail(m, COREG_EINVAL, "lags: null array or empty axis");
    const int n1 = lags->n_crval1, n2 = lags->n_crval2, world = m->n;
    const long long inner = (long long)lags->n_cdelt1 * lags->n_cdelt2 * lags->n_crota;
    const long long n_lags = (long long)n1 * n2 * inner;
    if (!corr_out && n_lags > 0) return mfail(m, COREG_EINVAL, "corr_out is null");
    int mode = multi_lag_sharding(n1, n2, inner, world);
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
    const bool rccl = m->use_rccl;
    m->collective = rccl ? "rccl" : "host-copy";

    if (mode == MULTI_POINTS) {
        // every device sweeps ALL lag-points over its share of the grid; the six sums per lag slot are added over the
        // devices (all-reduce), device 0 evaluates the coefficients
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
        if (rccl && n6 > 0) {
            ncclResult_t e = RcclApi::get().GroupStart();
            for (int k = 0; k < world && e == ncclSuccess; ++k)
                e = RcclApi::get().AllReduce(m->h[k]->sums.p, m->h[k]->sums.p, (size_t)n6, ncclDouble, ncclSum, m->comms[k],
                                             m->h[k]->stream);
            const ncclResult_t e2 = RcclApi::get().GroupEnd();
            if (e != ncclSuccess || e2 != ncclSuccess) rc = mfail(m, COREG_EHIP, "RCCL all-reduce failed");
            if (rc == COREG_OK) {
                m->w[0]->post([&] { rc = coreg_finalize_sums(m->h[0], m->h[0]->sums.as<double>(), 1, corr_out, 0); });
                m->w[0]->wait();
                if (rc != COREG_OK) mfail(m, rc, coreg_last_error(m->h[0]));
            }
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
        // back to unsharded contexts, and every stream drained before the caller's buffers go
        const int rc2 = multi_run(m, [&](int k) {
            RETCHK(coreg_set_option(m->h[k], "shard_world", 1));
            return coreg_synchronize(m->h[k]);
        });
        return rc != COREG_OK ? rc : rc2;
    }

    // ---- lag sharding: blocks of the (CRVAL1, CRVAL2) plane, or contiguous slices of the raveled index
    long long chunk;
    if (mode == MULTI_BLOCKS) {
        int g1, g2;
        multi_block_grid(n1, n2, world, &g1, &g2);
        chunk = (long long)((n1 + g1 - 1) / g1) * ((n2 + g2 - 1) / g2) * inner;
    } else {
        chunk = (n_lags + world - 1) / world;
    }
    if (!rccl && m->host_gather.reserve((size_t)chunk * world * sizeof(double)) != hipSuccess)
        return mfail(m, COREG_ENOMEM, "hipHostMalloc (gather buffer) failed");
    std::vector<long long> n_mine(world, 0);
    RETCHK(multi_run(m, [&](int k) {
        coreg_handle* h = m->h[k];
        RETCHK(bind_device(h));
        HIPCHK(m->blk[k].reserve((size_t)chunk * sizeof(double)));
        if (rccl) HIPCHK(m->gat[k].reserve((size_t)chunk * world * sizeof(double)));
        coreg_lags sub = *lags;
        int64_t lo = 0, hi = 0;
        if (mode == MULTI_BLOCKS) {
            int b[4];
            multi_block_bounds(n1, n2, world, k, b);
            sub.crval1 = lags->crval1 + b[0];
            sub.n_crval1 = b[1] - b[0];
            sub.crval2 = lags->crval2 + b[2];
            sub.n_crval2 = b[3] - b[2];
            hi = (int64_t)sub.n_crval1 * sub.n_crval2 * inner;
        } else {
            lo = std::min<long long>((long long)k * chunk, n_lags);
            hi = std::min<long long>((long long)(k + 1) * chunk, n_lags);
        }
        n_mine[k] = hi - lo;
        if (hi > lo) RETCHK(launch(k, &sub, lo, hi, m->blk[k].as<double>()));
        if (!rccl) {
            // this device's block straight to its place in the host buffer (asynchronous; drained below)
            if (hi > lo)
                HIPCHK(hipMemcpyAsync((double*)m->host_gather.p + (size_t)k * chunk, m->blk[k].p,
                                      (size_t)(hi - lo) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        return COREG_OK;
    }));
    const double* gathered = (const double*)m->host_gather.p;
    if (rccl) {
        // THE collective: one all-gather of `chunk` doubles per device, all devices in one group, each on the stream its
        // sweep was enqueued on
        ncclResult_t e = RcclApi::get().GroupStart();
        for (int k = 0; k < world && e == ncclSuccess; ++k)
            e = RcclApi::get().AllGather(m->blk[k].p, m->gat[k].p, (size_t)chunk, ncclDouble, m->comms[k], m->h[k]->stream);
        const ncclResult_t e2 = RcclApi::get().GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess) return mfail(m, COREG_EHIP, "RCCL all-gather failed");
        if (m->host_gather.reserve((size_t)chunk * world * sizeof(double)) != hipSuccess)
            return mfail(m, COREG_ENOMEM, "hipHostMalloc (gather buffer) failed");
        RETCHK(multi_run(m, [&](int k) {
            coreg_handle* h = m->h[k];
            RETCHK(bind_device(h));
            if (k == 0)  // every device holds the whole map; device 0 hands it to the host
                HIPCHK(hipMemcpyAsync(m->host_gather.p, m->gat[0].p, (size_t)chunk * world * sizeof(double),
                                      hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            return COREG_OK;
        }));
        gathered = (const double*)m->host_gather.p;
    }
    // gathered chunks -> C-order raveled map
    if (mode == MULTI_SLICES) {
        for (int k = 0; k < world; ++k)
            if (n_mine[k] > 0)
                std::memcpy(corr_out + (size_t)k * chunk, gathered + (size_t)k * chunk, (size_t)n_mine[k] * sizeof(double));
    } else {
        for (int k = 0; k < world; ++k) {
            int b[4];
            multi_block_bounds(n1, n2, world, k, b);
            const int w2 = b[3] - b[2];
            for (int i1 = b[0]; i1 < b[1]; ++i1)
                for (int i2 = b[2]; i2 < b[3]; ++i2)
                    std::memcpy(corr_out + ((size_t)i1 * n2 + i2) * inner,
                                gathered + (size_t)k * chunk + ((size_t)(i1 - b[0]) * w2 + (i2 - b[2])) * inner,
                                (size_t)inner * sizeof(double));
        }
    }
    return COREG_OK;
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

int coreg_multi_plan(int32_t n_crval1, int32_t n_crval2, int64_t n_inner, int32_t world, int32_t* mode, int32_t* g1,
                     int32_t* g2) {
    if (n_crval1 < 1 || n_crval2 < 1 || n_inner < 1 || world < 1 || !mode || !g1 || !g2) return COREG_EINVAL;
    *mode = multi_lag_sharding(n_crval1, n_crval2, n_inner, world);
    int a, b;
    multi_block_grid(n_crval1, n_crval2, world, &a, &b);
    *g1 = a;
    *g2 = b;
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
    // one communicator per device, created together (ncclCommInitAll: the single-process form)
    const char* force = std::getenv("COREG_MULTI_FORCE_RCCL");
    m->force_collective = m->n == 1 && force && std::atoi(force) == 1;
    if ((m->n > 1 || m->force_collective) && distinct && RcclApi::get().ok()) {
        m->comms.assign(m->n, nullptr);
        if (RcclApi::get().CommInitAll(m->comms.data(), m->n, m->devices.data()) == ncclSuccess) m->use_rccl = true;
        else m->comms.clear();
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
        });
        m->w[k]->wait();
    }
    if (m->use_rccl)
        for (ncclComm_t c : m->comms)
            if (c) (void)RcclApi::get().CommDestroy(c);
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
int coreg_multi_last_mode(const coreg_multi* m) { return m ? m->last_mode : 0; }

int coreg_multi_set_option(coreg_multi* m, const char* name, int64_t value) {
    if (!m) return COREG_EINVAL;
    return multi_run(m, [&](int k) { return coreg_set_option(m->h[k], name, value); });
}

int coreg_multi_set_small(coreg_multi* m, const void* img, int dtype, int32_t ny, int32_t nx) {
    if (!m) return COREG_EINVAL;
    if (!img || ny < 1 || nx < 1 || (dtype != COREG_F32 && dtype != COREG_F64))
        return mfail(m, COREG_EINVAL, "multi_set_small: bad argument");
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

int coreg_multi_sweep_carrington(coreg_multi* m, const coreg_wcs2d* hdr_small, const coreg_carr_grid* grid, double solar_r,
                                 const coreg_lags* lags, int order, int method, int cdelt_semantics, double* corr_out) {
    return multi_sweep(m, lags, corr_out, [&](int k, const coreg_lags* l, int64_t lo, int64_t hi, double* out_dev) {
        return out_dev ? coreg_sweep_carrington(m->h[k], hdr_small, grid, solar_r, l, order, method, cdelt_semantics, lo,
                                                hi, out_dev, 1)
                       : coreg_sweep_carrington(m->h[k], hdr_small, grid, solar_r, l, order, method, cdelt_semantics, lo,
                                                hi, corr_out, 0);
    });
}

int coreg_multi_sweep_helioprojective(coreg_multi* m, const coreg_wcs2d* hdr_target, const coreg_wcs2d* hdr_small,
                                      const coreg_lags* lags, int order, int method, int cdelt_semantics, double* corr_out) {
    return multi_sweep(m, lags, corr_out, [&](int k, const coreg_lags* l, int64_t lo, int64_t hi, double* out_dev) {
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
