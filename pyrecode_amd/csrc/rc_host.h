// rc_host.h - host-side plumbing shared by the translation units behind the C ABI (rc_api.hip: contexts and seam 1;
// rc_reader.hip: the batched reader; rc_codec_api.hip: the stateless codec seams): status / error text, device guard, device
// scratch helpers, and the per-GPU utility context the stateless entry points share.  The few globals are defined in rc_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <algorithm>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>
#include <sched.h>
#include <unistd.h>
#include <vector>

#include "../../include/recode_hip.h"
#include "rc_expand.h"
#include "rc_launch.h"
#include "rc_zstd_block.h"
#include "rc_zstd_dec.h"

#define RC_EXPORT extern "C" __attribute__((visibility("default")))

extern thread_local std::string g_last_error;   // rc_last_error(): one per thread, whichever translation unit failed

namespace {
int fail(int code, const char *what)
{
    g_last_error = what ? what : "";
    return code;
}
int hip_fail(hipError_t e, const char *where)
{
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    if (e == hipErrorOutOfMemory) {   // the workspace does not fit this GPU's free memory: a status of its own, and no sticky HIP error left behind
        (void)hipGetLastError();
        return RC_ERR_WORKSPACE;
    }
    return RC_ERR_DEVICE;
}
#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);     \
    } while (0)

// true when p is memory the GPU kernels can dereference (device or managed); false for ordinary host memory
bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host pointer: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

template <class T>
int ensure(T *&buf, uint64_t &cap, uint64_t need)
{
    if (need <= cap && buf) return RC_OK;
    if (buf) HIP_TRY(hipFree(buf));
    buf = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), need ? need : 16));
    cap = need;
    return RC_OK;
}

int copy_out(void *dst, const void *src_dev, uint64_t bytes, hipStream_t s)
{
    if (!bytes) return RC_OK;
    HIP_TRY(hipMemcpyAsync(dst, src_dev, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    return RC_OK;
}

// CPUs this process may REALLY use: the scheduler affinity capped by the cgroup's CPU bandwidth quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us) - the GPU boxes show 256 cores and grant 16; a pool sized from hardware_concurrency() spends the
// quota early in every accounting period and is throttled for the rest of it (pyrecode_amd/misc.py::effective_cpus is the Python twin).
uint32_t usable_cpus()
{
    static const uint32_t n = [] {
        uint32_t vis = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) vis = (uint32_t)CPU_COUNT(&set);
        if (!vis) vis = std::max(1u, std::thread::hardware_concurrency());
        double quota = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            long long period = 0;
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / (double)period;
            fclose(f);
        } else {
            long long q = -1, period = 0;
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
            if (q > 0 && period > 0) quota = (double)q / (double)period;
        }
        if (quota > 0) vis = std::max(1u, std::min(vis, (uint32_t)(quota + 0.999)));
        return vis;
    }();
    return n;
}

// Every entry point runs on its ctx's (or the utility context's) device and puts the caller's current device back on
// return: in a one-process-per-GPU job the thread's current device belongs to the caller (torch, RCCL), not to this library.
struct DeviceGuard {
    int prev = -1;
    bool moved = false;
    hipError_t enter(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev == dev) return hipSuccess;
        hipError_t e = hipSetDevice(dev);
        moved = e == hipSuccess && prev >= 0;
        return e;
    }
    ~DeviceGuard() { if (moved) (void)hipSetDevice(prev); }
};
#define RC_ON_DEVICE(dev) DeviceGuard dev_guard_; HIP_TRY(dev_guard_.enter(dev))
}  // namespace

// ---- utility contexts for the stateless seams (2 and 3) ----------------------------------------------------------
// One per GPU, created on first use.  A call runs on RC_DEVICE (env) when that is set, otherwise on the CALLER'S CURRENT
// device - in a one-process-per-GPU job that is the rank's own GPU - and leaves the current device as it found it.
// (types with one shared instance each - defined in rc_api.hip - live at namespace scope; the helpers around them are per-TU)
// A few worker threads that stay around between calls (rc_expand_frames indexes its frames on them: starting 15 threads per call
// cost more than the indexing itself - 0.5 of 0.7 ms for 64 frames).  run(n, fn) calls fn(0..n-1), fn(0) on the calling thread, and
// returns when all are done; runs are serialised (callers on different devices share the pool).  Never destroyed: the workers sleep
// on a condition variable and end with the process.  A forked child starts its own.
struct WorkerPool {
    std::mutex mu, run_mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(uint32_t)> fn;
    uint64_t generation = 0;
    uint32_t want = 0, done = 0;
    int started = 0;
    pid_t pid = 0;
    void worker(uint32_t id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(uint32_t)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return generation != seen; });
                seen = generation;
                if (id >= want) continue;
                f = fn;
            }
            f(id);
            {
                std::lock_guard<std::mutex> lk(mu);
                ++done;
            }
            cv_done.notify_one();
        }
    }
    void run(uint32_t n, const std::function<void(uint32_t)> &f)
    {
        if (n <= 1) { if (n) f(0); return; }
        std::lock_guard<std::mutex> one_run(run_mu);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (pid != getpid()) { started = 0; pid = getpid(); }   // (after a fork the parent's workers do not exist here)
            for (; started + 1 < (int)n; ++started) std::thread(&WorkerPool::worker, this, (uint32_t)started + 1).detach();
            fn = f;
            want = n;
            done = 1;   // id 0 runs here
            ++generation;
        }
        cv_go.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done >= want; });
    }
};
extern WorkerPool *g_pool;

// Growable array in page-locked host memory (a hipMemcpyAsync from it is a real asynchronous copy; capacity is kept).
template <class T>
struct PinnedVec {
    T *p = nullptr; size_t n = 0, cap = 0;
    bool ok = true;                     // false: an allocation failed (checked by the caller after the indexing threads have joined)
    void clear() { n = 0; ok = true; }
    size_t size() const { return n; }
    const T *data() const { return p; }
    bool grow(size_t nc)
    {
        T *q = nullptr;
        if (hipHostMalloc((void **)&q, nc * sizeof(T), hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); ok = false; return false; }
        if (n) memcpy(q, p, n * sizeof(T));
        if (p) (void)hipHostFree(p);
        p = q; cap = nc;
        return true;
    }
    // room for `want` entries up front: every doubling is a page-locked allocation (about a millisecond), and a reader's first
    // batches otherwise pay five or six of them per indexing thread
    void reserve(size_t want) { if (want > cap) (void)grow(want); }
    void push_back(const T &v)
    {
        if (n == cap && !grow(cap ? cap * 2 : 8192)) return;
        p[n++] = v;
    }
};
constexpr int RC_READ_THREADS = 16;
constexpr int RC_READ_SLOTS = 2;

// Everything one batch of the batched reader owns while it is in flight (rc_expand_frames uses slot 0; rc_expand_frames_submit /
// _wait alternate between the slots, so that the host walk and copy-in of one batch run while the device decodes the other).
// Kept between calls: no allocation and no first-touch page faults in steady state.
struct ReadRes {
    hipStream_t stream = nullptr, stream2 = nullptr;       // the two streams' decoders run side by side
    hipEvent_t ev_a = nullptr, ev_b = nullptr, done = nullptr;
    uint8_t *x[10] = {}; uint64_t x_cap[10] = {};          // device: data, decoded streams, -, head, -, counters, staged triplets
    PinnedVec<rc::ZdBlock> rd_bm[RC_READ_THREADS], rd_pv[RC_READ_THREADS], rd_raw[RC_READ_THREADS];   // per indexing thread, page-locked
    std::vector<rc::ZdBlock> rd_tmp[RC_READ_THREADS];
    PinnedVec<uint32_t> rd_off[RC_READ_THREADS];           // compact lists of uniform binary-map streams: one header offset per block (k_bitmap_decode_c)
    uint8_t *rd_head = nullptr; uint64_t rd_head_cap = 0;  // page-locked: decoding tables + per-frame index arrays
    uint64_t *h_res = nullptr; uint64_t h_res_cap = 0;     // page-locked: nnz prefix (n + 1) and the error word, as the device left them
    uint8_t *h_blob = nullptr; uint64_t h_blob_cap = 0;    // page-locked: host copy of a DEVICE-resident input, for the header walk
    // a submitted batch waiting for its rc_expand_frames_wait
    bool pending = false;
    uint32_t n = 0, level = 0, bit_depth = 0;
    uint64_t cap = 0;
    std::vector<uint32_t> pv_bytes;
};

struct Util {
    std::mutex mu;
    int device = -1;
    hipStream_t stream = nullptr;
    uint8_t *a = nullptr; uint64_t a_cap = 0;   // input 1
    uint8_t *b = nullptr; uint64_t b_cap = 0;   // input 2
    uint8_t *o = nullptr; uint64_t o_cap = 0;   // output
    uint8_t *w = nullptr; uint64_t w_cap = 0;   // work
    uint64_t *h_scalar = nullptr;               // pinned
    void *ztab = nullptr;                       // zstd FSE tables
    uint8_t *x[10] = {}; uint64_t x_cap[10] = {};   // rc_expand_frames: data, bitmaps, values, tables, block lists, counters
    void *zd_predef = nullptr;                  // predefined zstd decoding tables
    ReadRes rr[RC_READ_SLOTS + 1];              // the submit / wait form's two slots, then the synchronous rc_expand_frames' own:
                                                // a synchronous call (e.g. the reader's fallback for ONE batch) never meets a queued batch
};
constexpr int RC_MAX_DEV = 64;
extern Util g_utils[RC_MAX_DEV];
extern thread_local Util *t_util;
namespace {
#define g_util (*t_util)

struct UtilScope {
    DeviceGuard guard;
    std::unique_lock<std::mutex> lock;
    int enter()
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
            (void)hipGetLastError();
            return fail(RC_ERR_DEVICE, "no HIP device visible (this library has no CPU path)");
        }
        int dev = 0;
        const char *env = getenv("RC_DEVICE");
        if (env) dev = atoi(env);
        else if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        if (dev < 0 || dev >= ndev || dev >= RC_MAX_DEV) return fail(RC_ERR_BAD_ARG, "RC_DEVICE out of range");
        t_util = &g_utils[dev];
        lock = std::unique_lock<std::mutex>(t_util->mu);
        HIP_TRY(guard.enter(dev));
        if (t_util->device < 0) {
            HIP_TRY(hipStreamCreateWithFlags(&t_util->stream, hipStreamNonBlocking));
            HIP_TRY(hipHostMalloc((void **)&t_util->h_scalar, 64, hipHostMallocDefault));
            t_util->device = dev;
        }
        return RC_OK;
    }
};

// device-visible view of caller memory: the pointer itself, or a staged copy in `buf`
template <class T>
int stage_in(const T *src, uint64_t bytes, uint8_t *&buf, uint64_t &cap, const T *&dev, uint64_t pad = 0)
{
    if (is_device_ptr(src) && pad == 0) {
        dev = src;
        return RC_OK;
    }
    int r = ensure(buf, cap, bytes + pad);
    if (r != RC_OK) return r;
    if (pad) HIP_TRY(hipMemsetAsync(buf + bytes, 0, pad, g_util.stream));
    if (bytes)
        HIP_TRY(hipMemcpyAsync(buf, src, bytes, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               g_util.stream));
    dev = reinterpret_cast<const T *>(buf);
    return RC_OK;
}
}  // namespace
