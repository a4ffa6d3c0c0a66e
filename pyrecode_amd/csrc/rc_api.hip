// rc_api.hip - the C ABI of librecode_hip.so (include/recode_hip.h): contexts, staging, entry points.
// No CPU implementation lives here: every compute entry point runs HIP kernels or returns RC_ERR_DEVICE.
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
#include <unistd.h>
#include <vector>

#include "../../include/recode_hip.h"
#include "rc_expand.h"
#include "rc_launch.h"
#include "rc_zstd_block.h"
#include "rc_zstd_dec.h"

#define RC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *what)
{
    g_last_error = what ? what : "";
    return code;
}
int hip_fail(hipError_t e, const char *where)
{
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
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

struct rc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint32_t nx = 0, ny = 0, depth = 0, level = 0, op_mode = 0, scheme = 0, clevel = 0, max_batch = 0;
    uint32_t emit = 0;  // 0: mode-0 record pieces, 2: LZ4 frames
    // Per-batch scratch exists twice: batch i reduces into sets[i & 1] on `stream`; its scans / layout / assembly (small,
    // latency-bound kernels that leave most of the GPU idle) run on `pstream` and may overlap the next batch's reduce
    // kernel (rc_ctx_set_pipelined).  `sc` is the set of the most recent batch (same geometry and threshold in both).
    rc::Scratch sc, sets[2];
    int cur = 0;                          // set the NEXT batch uses
    int last = 0;                         // set of the most recent batch
    hipStream_t pstream = nullptr;        // carries everything behind the reduce kernel: one of the two below
    hipStream_t pstream_all = nullptr, pstream_masked = nullptr;
    hipEvent_t ev_red[2] = {}, ev_post[2] = {}, ev_in[2] = {};
    bool post_pending[2] = {false, false};
    bool pipelined = false;
    bool thr_set = false;
    bool keep_bitmap = true;  // also store the raw binary maps when a device codec is active (rc_get_binary_map)
    uint32_t last_n = 0;
    // staging for host callers
    uint16_t *d_frames = nullptr; uint64_t d_frames_cap = 0;
    uint8_t *d_out = nullptr;     uint64_t d_out_cap = 0;
    uint16_t *d_dark = nullptr;   uint64_t d_dark_cap = 0;
    uint64_t *d_rec_off = nullptr;
    uint32_t *d_md = nullptr;
    void *d_ztab = nullptr;               // zstd FSE tables (emit == 1)
    // modelled zstd (compression_level >= 1): tables fitted to a sample of the ctx's first batch (rc_zstd_model.h)
    bool modelled = false, model_ready = false;
    rc::ZstdModel *d_model = nullptr, *h_model = nullptr;
    rc::ZstdSample *d_sample = nullptr, *h_sample = nullptr;
    rc::L2Work l2;                        // level 2 workspace
    uint32_t l2_sum = 0;                  // L2_statistics: 0/1 max, 2 sum
    rc::BatchStatus *h_status = nullptr;  // pinned: [0] most recent batch, [1] first failed batch since the last sync
    rc::BatchStatus *d_first_err = nullptr;
    uint32_t batch_seq = 0;               // batches enqueued since the last rc_ctx_sync
    hipEvent_t ev[5] = {};
    float stage_ms[5] = {};
    // optional per-enqueue stage events for the asynchronous path (rc_ctx_set_profiling)
    bool profiling = false;
    bool profile_all = getenv("RC_PROFILE_ALL_STAGES") != nullptr;
    // host streaming form (rc_pipe_*): per slot device buffers, pinned metadata, events
    struct PipeSlot {
        uint16_t *d_in = nullptr;
        uint8_t *d_out = nullptr;
        uint64_t *d_rec = nullptr, *h_rec = nullptr;
        uint32_t *d_md = nullptr, *h_md = nullptr;
        rc::BatchStatus *h_stat = nullptr;
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_fetch = nullptr, ev_val = nullptr;
        uint32_t *d_val = nullptr, *h_val = nullptr;   // validation frames: component counts of the ROI (0xFFFFFFFF: not a validation frame)
        bool has_val = false;
        uint32_t n = 0;
        bool zero_copy = false;
        int state = 0;   // 0 free, 1 submitted, 2 result taken, 3 fetching
    } pipe[RC_PIPE_SLOTS];
    hipStream_t copy_stream = nullptr, d2h_stream = nullptr;
    uint32_t val_gap = 0, val_x0 = 0, val_y0 = 0, val_w = 0, val_h = 0;   // rc_ctx_set_validation
    std::vector<hipEvent_t> prof_ev;   // 5 events per enqueued batch, in enqueue order
    size_t prof_used = 0;              // events consumed since the last rc_ctx_sync
    double prof_sum_ms[5] = {};
    uint64_t prof_batches = 0;
};

// ---- library ---------------------------------------------------------------------------------------------
RC_EXPORT int rc_abi_version(void) { return RC_ABI_VERSION; }

RC_EXPORT const char *rc_strerror(int status)
{
    switch (status) {
    case RC_OK: return "ok";
    case RC_ERR_BAD_ARG: return "bad argument";
    case RC_ERR_OUT_TOO_SMALL: return "output buffer too small";
    case RC_ERR_DEVICE: return "GPU / HIP error";
    case RC_ERR_UNSUPPORTED: return "not implemented on device";
    case RC_ERR_RECORD_TOO_LARGE: return "Buffer size smaller than compressed data size";
    case RC_ERR_CORRUPT: return "corrupt input stream";
    case RC_ERR_WORKSPACE: return "level-2 workspace exceeded (too many foreground pixels in the batch)";
    default: return "unknown status";
    }
}
RC_EXPORT const char *rc_last_error(void) { return g_last_error.c_str(); }

RC_EXPORT int rc_device_count(int *count)
{
    if (!count) return fail(RC_ERR_BAD_ARG, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return RC_OK;
}
RC_EXPORT int rc_scheme_on_device(uint32_t scheme)
{
    return (scheme == RC_SCHEME_LZ4 || scheme == RC_SCHEME_ZSTD || scheme == RC_SCHEME_BLOSC_LZ4) ? 1 : 0;
}

// ---- seam 1 --------------------------------------------------------------------------------------------------
static int alloc_set(rc_ctx *c, rc::Scratch &sc)
{
    using namespace rc;
    const uint64_t B = c->max_batch, T = sc.ntiles;
    HIP_TRY(hipMalloc((void **)&sc.bitmap, B * sc.nb_stride + 64));  // + slack: wave_copy reads <= 4 B past a tile
    HIP_TRY(hipMalloc((void **)&sc.tile_cnt, B * T * 4));
    HIP_TRY(hipMalloc((void **)&sc.tile_off, B * T * 4));
    HIP_TRY(hipMalloc((void **)&sc.tile_next, B * T * 4));
    HIP_TRY(hipMalloc((void **)&sc.frame_nnz, B * 4));
    HIP_TRY(hipMalloc((void **)&sc.frame_cbytes, B * 4));
    HIP_TRY(hipMalloc((void **)&sc.scan_part, B * ((T + 4095) / 4096) * 32));
    HIP_TRY(hipMalloc((void **)&sc.status, sizeof(BatchStatus)));
    if (c->level != 3) HIP_TRY(hipMalloc((void **)&sc.pix_slots, B * T * TILE_PX * 2 + 64));
    if (c->emit != 0) {
        HIP_TRY(hipMalloc((void **)&sc.blk_slots, B * T * BLK_SLOT + 64));
        HIP_TRY(hipMalloc((void **)&sc.blk_size, B * T * 4));
        HIP_TRY(hipMalloc((void **)&sc.blk_off, B * T * 4));
    }
    if (c->emit == RC_SCHEME_ZSTD && c->clevel != 0 && c->level == 1) {   // modelled zstd: Huffman stage of the residual stream
        sc.pixraw_stride = ((sc.N * 2 + 15) & ~15ull) + 32;
        sc.nchunk_max = (uint32_t)((sc.N * 2 + PIX_CHUNK - 1) / PIX_CHUNK) + 1;
        HIP_TRY(hipMalloc((void **)&sc.pixraw, B * sc.pixraw_stride + 64));
        HIP_TRY(hipMalloc((void **)&sc.pix_chunks, B * (uint64_t)sc.nchunk_max * PIX_SLOT + 64));
        HIP_TRY(hipMalloc((void **)&sc.chunk_size, B * (uint64_t)sc.nchunk_max * 4));
        HIP_TRY(hipMalloc((void **)&sc.chunk_off, B * (uint64_t)sc.nchunk_max * 4));
        HIP_TRY(hipMalloc((void **)&sc.frame_pbytes, B * 4));
        HIP_TRY(hipMemset(sc.frame_pbytes, 0, B * 4));
    }
    HIP_TRY(hipMemset(sc.frame_nnz, 0, B * 4));
    HIP_TRY(hipMemset(sc.frame_cbytes, 0, B * 4));
    HIP_TRY(hipMemset(sc.status, 0, sizeof(BatchStatus)));
    return RC_OK;
}

// (re)allocate the level-2 workspace for `cap` set pixels per batch
static int l2_alloc(rc_ctx *c, uint64_t cap)
{
    using namespace rc;
    L2Work &w = c->l2;
    void *old[] = {w.pos, w.val, w.parent, w.stat};
    for (void *b : old) if (b) HIP_TRY(hipFree(b));
    w.pos = nullptr; w.val = nullptr; w.parent = nullptr; w.stat = nullptr;
    const uint64_t all = (uint64_t)c->max_batch * c->sc.N;
    w.cap = std::min<uint64_t>(std::min<uint64_t>(cap, all), 0xFFFFFFF0ull);
    HIP_TRY(hipMalloc((void **)&w.pos, w.cap * 4));
    HIP_TRY(hipMalloc((void **)&w.val, w.cap * 2));
    HIP_TRY(hipMalloc((void **)&w.parent, w.cap * 4));
    HIP_TRY(hipMalloc((void **)&w.stat, w.cap * 4));
    if (!w.word_rank) {
        w.words_per_frame = (uint64_t)c->sc.ntiles * (TILE_PX / 64);
        HIP_TRY(hipMalloc((void **)&w.word_rank, (uint64_t)c->max_batch * w.words_per_frame * 4));
        HIP_TRY(hipMalloc((void **)&w.frame_base, ((uint64_t)c->max_batch + 1) * 8));
    }
    return RC_OK;
}

static int ctx_alloc(rc_ctx *c)
{
    using namespace rc;
    const uint64_t B = c->max_batch;
    RC_ON_DEVICE(c->device);
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    // (a high-priority second-stage stream was measured: no gain with LZ4 or zstd, 2 % slower at 11520x8184 - tools/ab_bench.sh)
    HIP_TRY(hipStreamCreateWithFlags(&c->pstream_all, hipStreamNonBlocking));
    {
        // Experiment knob, off by default.  In pipelined mode the second stage runs next to the following batch's reduce
        // kernel; its waves (80 VGPRs, latency bound) settle on every SIMD and push out one of the three reduce waves
        // there (168 VGPRs each).  RC_PSTREAM_CUS=n confines the second stage to the first n CU-mask bits.  Measured on
        // bench.py, same box: LZ4 123.7 k frames/s unmasked, 126.2 k with n = 104 (96: 125.6-127.6 k, 128: 128.5 k on a
        // faster box, 64: no gain, spread-out masks: worse) - but zstd, whose second stage also carries the FSE kernel,
        // drops from 112 k to 103 k: the confined stage becomes the longer one.  +2 % on one codec does not pay for that.
        const char *e = getenv("RC_PSTREAM_CUS");
        const int ncu = e ? atoi(e) : 0;
        c->pstream_masked = nullptr;
        if (ncu > 0 && ncu < 256) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&c->pstream_masked, 8, mask) != hipSuccess) {
                (void)hipGetLastError();
                c->pstream_masked = nullptr;  // not available: pipelined mode uses the unmasked stream
            }
        }
    }
    c->pstream = c->pstream_all;
    c->stream = c->own_stream;
    HIP_TRY(hipMalloc((void **)&c->sc.thr, c->sc.N * 2));
    HIP_TRY(hipMalloc((void **)&c->d_first_err, sizeof(BatchStatus)));
    HIP_TRY(hipMemset(c->d_first_err, 0, sizeof(BatchStatus)));
    c->sc.first_err = c->d_first_err;
    for (Scratch &set : c->sets) {
        set = c->sc;  // geometry + the shared threshold
        int r = alloc_set(c, set);
        if (r != RC_OK) return r;
    }
    c->sc = c->sets[0];
    if (c->level == 2) {
        // compact-pixel workspace: room for EVERY pixel of a batch (14 bytes each - 7.5 GB for 32 frames of 4096^2, nothing
        // next to 288 GB), so that no batch can exceed it; should that allocation fail, 12.5 % mean foreground, grown on demand
        // by the synchronous entry point (RC_ERR_WORKSPACE from the asynchronous ones)
        int r = getenv("RC_L2_SMALL_WORKSPACE") ? RC_ERR_DEVICE : l2_alloc(c, B * c->sc.N);   // (the env switch: tests of the growth path)
        if (r != RC_OK) {
            (void)hipGetLastError();
            r = l2_alloc(c, std::max<uint64_t>(B * c->sc.N / 8, 1ull << 16));
        }
        if (r != RC_OK) return r;
    }
    if (c->emit == RC_SCHEME_ZSTD) {
        std::vector<uint8_t> tab(zstd_tables_bytes());
        zstd_tables_host(tab.data());
        HIP_TRY(hipMalloc(&c->d_ztab, tab.size()));
        HIP_TRY(hipMemcpy(c->d_ztab, tab.data(), tab.size(), hipMemcpyHostToDevice));
        // compression_level 0 = the fast encoder (raw literals, predefined tables, stored residuals); any other level = the
        // modelled one.  (The reference hands the level to libzstd, recode_writer.py:175-178; the device encoders have these two.)
        c->modelled = c->clevel != 0;
        if (c->modelled) {
            HIP_TRY(hipMalloc((void **)&c->d_model, sizeof(ZstdModel)));
            HIP_TRY(hipMalloc((void **)&c->d_sample, sizeof(ZstdSample)));
            HIP_TRY(hipHostMalloc((void **)&c->h_model, sizeof(ZstdModel), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&c->h_sample, sizeof(ZstdSample), hipHostMallocDefault));
        }
    }
    HIP_TRY(hipMalloc((void **)&c->d_rec_off, (B + 1) * 8));
    HIP_TRY(hipMalloc((void **)&c->d_md, B * 3 * 4));
    HIP_TRY(hipHostMalloc((void **)&c->h_status, 2 * sizeof(BatchStatus), hipHostMallocDefault));
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto &e : c->ev_red) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : c->ev_post) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : c->ev_in) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return RC_OK;
}

RC_EXPORT rc_ctx *rc_ctx_create(uint32_t nx, uint32_t ny, uint32_t src_bit_depth, uint32_t reduction_level,
                                uint32_t op_mode, uint32_t scheme, uint32_t clevel, int device_id, uint32_t max_batch,
                                int *status)
{
    int dummy;
    if (!status) status = &dummy;
    *status = RC_OK;
    if (nx == 0 || ny == 0 || max_batch == 0 || op_mode > 1) {
        *status = fail(RC_ERR_BAD_ARG, "nx, ny, max_batch must be > 0 and op_mode 0 or 1");
        return nullptr;
    }
    if ((uint64_t)nx * ny >= (1ull << 32)) {
        *status = fail(RC_ERR_BAD_ARG, "nx*ny must be < 2^32");
        return nullptr;
    }
    if (reduction_level < 1 || reduction_level > 3) {
        *status = fail(RC_ERR_UNSUPPORTED, "reduction_level 4 (centroiding) is not implemented on device");
        return nullptr;
    }
    if (src_bit_depth < 9 || src_bit_depth > 16) {
        *status = fail(RC_ERR_UNSUPPORTED, "source_bit_depth must be 9..16 (uint16 source frames)");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
        (void)hipGetLastError();
        *status = fail(RC_ERR_DEVICE, "no such HIP device (this library has no CPU path)");
        return nullptr;
    }
    rc_ctx *c = new (std::nothrow) rc_ctx();
    if (!c) {
        *status = fail(RC_ERR_DEVICE, "out of host memory");
        return nullptr;
    }
    c->device = device_id;
    c->nx = nx; c->ny = ny; c->depth = src_bit_depth; c->level = reduction_level; c->op_mode = op_mode;
    c->scheme = scheme; c->clevel = clevel; c->max_batch = max_batch;
    c->emit = (op_mode == 1 && rc_scheme_on_device(scheme)) ? scheme : 0;
    rc::Scratch &sc = c->sc;
    sc.N = (uint64_t)nx * ny;
    sc.ntiles = (uint32_t)((sc.N + rc::TILE_PX - 1) / rc::TILE_PX);
    sc.nb = (sc.N + 7) / 8;
    sc.nb_stride = (uint64_t)sc.ntiles * rc::TILE_BM;
    sc.max_batch = max_batch;
    int rcode = ctx_alloc(c);
    if (rcode != RC_OK) {
        *status = rcode;
        std::string keep = g_last_error;
        rc_ctx_destroy(c);
        g_last_error = keep;
        return nullptr;
    }
    return c;
}

RC_EXPORT int rc_ctx_destroy(rc_ctx *c)
{
    if (!c) return RC_OK;
    DeviceGuard dev_guard_;
    (void)dev_guard_.enter(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->pstream_all) (void)hipStreamSynchronize(c->pstream_all);
    if (c->pstream_masked) (void)hipStreamSynchronize(c->pstream_masked);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (rc::Scratch &sc : c->sets) {
        void *per_set[] = {sc.bitmap, sc.pix_slots, sc.tile_cnt, sc.tile_off, sc.tile_next, sc.blk_slots, sc.blk_size,
                           sc.blk_off, sc.frame_nnz, sc.frame_cbytes, sc.scan_part, sc.status, sc.pixraw, sc.pix_chunks, sc.chunk_size,
                           sc.chunk_off, sc.frame_pbytes};
        for (void *b : per_set)
            if (b) (void)hipFree(b);
    }
    for (auto &p : c->pipe) {
        void *dev[] = {p.d_in, p.d_out, p.d_rec, p.d_md, p.d_val};
        for (void *b : dev) if (b) (void)hipFree(b);
        void *host[] = {p.h_rec, p.h_md, p.h_stat, p.h_val};
        for (void *b : host) if (b) (void)hipHostFree(b);
        hipEvent_t evs[] = {p.ev_h2d, p.ev_done, p.ev_fetch, p.ev_val};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->d2h_stream) { (void)hipStreamSynchronize(c->d2h_stream); (void)hipStreamDestroy(c->d2h_stream); }
    if (c->h_model) (void)hipHostFree(c->h_model);
    if (c->h_sample) (void)hipHostFree(c->h_sample);
    void *bufs[] = {c->sc.thr, c->d_first_err, c->d_frames, c->d_out, c->d_dark, c->d_rec_off,
                    c->d_md, c->d_ztab, c->d_model, c->d_sample, c->l2.pos, c->l2.val, c->l2.parent, c->l2.stat, c->l2.word_rank, c->l2.frame_base};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    hipEvent_t sync_ev[] = {c->ev_red[0], c->ev_red[1], c->ev_post[0], c->ev_post[1], c->ev_in[0], c->ev_in[1]};
    for (hipEvent_t e : sync_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->pstream_all) (void)hipStreamDestroy(c->pstream_all);
    if (c->pstream_masked) (void)hipStreamDestroy(c->pstream_masked);
    if (c->h_status) (void)hipHostFree(c->h_status);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->prof_ev) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RC_OK;
}

RC_EXPORT int rc_ctx_set_stream(rc_ctx *c, void *hip_stream)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return RC_OK;
}

RC_EXPORT int rc_set_threshold(rc_ctx *c, const uint16_t *thr)
{
    if (!c || !thr) return fail(RC_ERR_BAD_ARG, "ctx / thr is NULL");
    RC_ON_DEVICE(c->device);
    HIP_TRY(hipMemcpyAsync(c->sc.thr, thr, c->sc.N * 2, is_device_ptr(thr) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->thr_set = true;
    return RC_OK;
}

RC_EXPORT int rc_set_dark(rc_ctx *c, const uint16_t *dark, int64_t epsilon)
{
    if (!c || !dark) return fail(RC_ERR_BAD_ARG, "ctx / dark is NULL");
    RC_ON_DEVICE(c->device);
    const uint16_t *src = dark;
    if (!is_device_ptr(dark)) {
        int r = ensure(c->d_dark, c->d_dark_cap, c->sc.N * 2);
        if (r != RC_OK) return r;
        HIP_TRY(hipMemcpyAsync(c->d_dark, dark, c->sc.N * 2, hipMemcpyHostToDevice, c->stream));
        src = c->d_dark;
    }
    rc::launch_threshold(src, epsilon, c->sc.N, c->sc.thr, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->thr_set = true;
    return RC_OK;
}

RC_EXPORT uint64_t rc_out_capacity(const rc_ctx *c, uint32_t n) { return c ? (uint64_t)n * c->sc.N * 2 : 0; }

RC_EXPORT uint32_t rc_md_fields(const rc_ctx *c)
{
    if (!c) return 0;
    const bool comp = c->emit != 0;
    if (c->level != 3) return comp ? 3 : 1;
    return comp ? 1 : 0;
}

// Modelled zstd: fit the ctx's tables to (up to two frames of) its first batch.  The sample is tokenized by the plain encoder
// into the scratch set the batch is about to use, k_zstd_sample turns the slots into histograms, the host builds the model
// (rc_zstd_model.h).  Synchronous, once per ctx; every later frame carries this model's descriptions.
static int fit_model(rc_ctx *c, const uint16_t *frames_dev, uint32_t n)
{
    using namespace rc;
    hipStream_t s = c->stream;
    const Scratch &sc = c->sets[c->cur];
    const uint32_t ns = n < 2 ? n : 2;
    for (int k = 0; k < 2; ++k)
        if (c->post_pending[k]) HIP_TRY(hipStreamWaitEvent(s, c->ev_post[k], 0));
    HIP_TRY(hipMemsetAsync(c->d_sample, 0, sizeof(ZstdSample), s));
    launch_reduce(sc, frames_dev, ns, c->level == 3 ? 3u : 1u, 1u, false, c->depth, s);
    launch_zstd_sample(sc, ns, c->level == 1, c->depth, c->d_sample, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_sample, c->d_sample, sizeof(ZstdSample), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    zstd_model_from_sample(c->h_sample, c->h_model);
    if (c->level != 1) c->h_model->valid &= ~2u;   // level 2 statistics / level 3: no residual-stream code
    {   // a residual stream the byte-wise code cannot shrink (bit-packed depths) is stored instead, in 128 KiB Raw blocks
        uint64_t bits = 0, total = 0;
        for (int v = 0; v < 256; ++v) { bits += (uint64_t)c->h_sample->pix[v] * (c->h_model->pix_code[v] >> 12); total += c->h_sample->pix[v]; }
        if (total == 0 || bits > total * 8 * 97 / 100) c->h_model->valid &= ~2u;
    }
    HIP_TRY(hipMemcpyAsync(c->d_model, c->h_model, sizeof(ZstdModel), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    const ZstdModel &M = *c->h_model;
    for (Scratch *set : {&c->sets[0], &c->sets[1], &c->sc}) {
        set->zm_model = c->d_model;
        set->zm_lit_code = c->d_model->lit_code;
        set->zm_valid = M.valid;
        set->zm_budget = zm_block_budget(M, BLK_SLOT);
        set->zm_seq_bits = (M.valid & 4u) ? M.seq.ll_log + M.seq.ml_log : 12u;
    }
    c->model_ready = true;
    return RC_OK;
}

static int enqueue_batch(rc_ctx *c, const uint16_t *frames_dev, uint32_t n, uint32_t first_frame_id, uint8_t *out_dev,
                         uint64_t out_cap, uint64_t *rec_off_dev, uint32_t *md_dev, bool timed)
{
    using namespace rc;
    hipStream_t s = c->stream;
    if (c->modelled && !c->model_ready) {
        int r = fit_model(c, frames_dev, n);
        if (r != RC_OK) return r;
    }
    RecordParams rp;
    rp.level = c->level == 3 ? 3u : 1u;  // level 2 records are framed exactly like level 1 (statistics in place of residuals)
    rp.emit = c->emit; rp.depth = c->depth; rp.first_frame_id = first_frame_id;
    rp.packed_slots = c->level == 1 ? 1u : 0u;
    rp.frame_bytes = c->sc.N * 2;
    hipEvent_t *ev = nullptr;
    if (timed) ev = c->ev;
    else if (c->profiling) {
        if (c->prof_used + 5 > c->prof_ev.size()) {
            for (int i = 0; i < 5; ++i) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                c->prof_ev.push_back(e);
            }
        }
        ev = c->prof_ev.data() + c->prof_used;
        c->prof_used += 5;
    }
    // Two streams, two scratch sets.  The reduce kernel runs on the ctx's stream `s` (behind whatever produced the frames
    // there); everything after it runs on `pstream`.  Default: `s` then waits for the batch's records, so stream order as
    // the caller sees it is the plain one.  Pipelined (rc_ctx_set_pipelined): `s` does not wait - the next batch's reduce
    // kernel overlaps this batch's scans / layout / assembly, and consumers order themselves with rc_ctx_wait_results.
    const int k = c->cur;
    c->cur ^= 1;
    c->last = k;
    c->sc = c->sets[k];
    const rc::Scratch &sc = c->sets[k];
    hipStream_t ps = c->pstream;
    if (c->post_pending[k]) HIP_TRY(hipStreamWaitEvent(s, c->ev_post[k], 0));  // the batch two calls ago has left this set
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));
    // every device codec's block encoder runs inside the reduce kernel (LZ4; blosc = bit-shuffle + LZ4; zstd: the
    // byte-parallel half - literals, sequence tokens - with the serial FSE half lane-per-block behind it)
    const bool fitted_seq = c->modelled && (c->h_model->valid & 4u);
    hipStream_t tail = nullptr;
    if (sc.N % TILE_PX) {   // a partial last tile: its small launch goes to the second-stage stream, behind "the frames are there"
        HIP_TRY(hipEventRecord(c->ev_in[k], s));
        HIP_TRY(hipStreamWaitEvent(ps, c->ev_in[k], 0));
        tail = ps;
    }
    // codec of the fused block encoder: 1 zstd fast, 3 zstd modelled, 2 LZ4 runs (compression_level 0), 4 LZ4 events (>= 1), 8 blosc
    const uint32_t codec = c->modelled ? 3u : (c->emit == RC_SCHEME_LZ4 && c->clevel != 0 ? 4u : c->emit);
    launch_reduce(sc, frames_dev, n, c->level, codec, c->keep_bitmap || c->emit == 0, c->depth, s, tail);
    // every event costs a few microseconds of stream time: the asynchronous path records only the ones it needs
    // (start, end of the reduce kernel, end of the batch) unless RC_PROFILE_ALL_STAGES is set
    const bool all_ev = ev && (timed || c->profile_all);
    // (one event behind the reduce kernel: every packet on this stream is a few microseconds between two reduce kernels)
    hipEvent_t red = ev ? ev[1] : c->ev_red[k];
    HIP_TRY(hipEventRecord(red, s));
    HIP_TRY(hipStreamWaitEvent(ps, red, 0));
    if (c->level == 2) {  // per-tile counts -> per-frame prefix, then connected components on the compacted pixels
        launch_scans(sc, n, true, false, ps);
        launch_l2(sc, c->l2, n, c->nx, c->l2_sum ? (1u << c->depth) - 1u : 0u, ps);
    }
#ifdef RC_DEV_SKIP   // development builds only (tools/build_def.sh): leave second-stage kernels out (WRONG records) to see what each costs the
                     // reduce kernel running next to it - bits: 1 FSE, 2 scans, 4 residual Huffman chain, 8 layout, 16 assemble, 32 gather
    static const unsigned skip = getenv("RC_DEV_SKIP_BITS") ? (unsigned)atoi(getenv("RC_DEV_SKIP_BITS")) : 0u;
#else
    constexpr unsigned skip = 0;
#endif
    if (c->emit == RC_SCHEME_ZSTD && !(skip & 1)) launch_zstd_fse(sc, n, fitted_seq ? (const void *)&c->d_model->seq : c->d_ztab, fitted_seq, ps);
    if (all_ev) HIP_TRY(hipEventRecord(ev[2], ps));
    if (!(skip & 2)) launch_scans(sc, n, c->level == 1, c->emit != 0, ps);  // (level 2: k_l2_emit has already described its value list)
    if (all_ev) HIP_TRY(hipEventRecord(ev[3], ps));
    // modelled zstd, level 1: the residual stream is laid out flat, Huffman-coded in chunks, and placed behind the bitmap stream
    // (rc_pix_huff.hip); its encoded size is part of the record layout
    const bool pix_huff = c->modelled && c->level == 1 && (c->h_model->valid & 2u) && sc.pixraw;
    if (pix_huff && !(skip & 4)) {
        rp.pix_mode = 1;
        launch_assemble(sc, rp, n, out_dev, rec_off_dev, c->batch_seq, ps);
        launch_pix_huff(sc, n, c->depth, ps);
        launch_pix_scan(sc, n, c->depth, ps);
        rp.pix_mode = 2;
    }
    if (!(skip & 8)) launch_layout(sc, rp, n, out_cap, rec_off_dev, md_dev, ps);
    if (!(skip & 16)) launch_assemble(sc, rp, n, out_dev, rec_off_dev, c->batch_seq, ps);
    ++c->batch_seq;
    if (pix_huff && !(skip & 32)) launch_pix_gather(sc, n, c->depth, 16, out_dev, rec_off_dev, ps);
    if (ev) HIP_TRY(hipEventRecord(ev[4], ps));
    HIP_TRY(hipEventRecord(c->ev_post[k], ps));
    c->post_pending[k] = true;
    if (!c->pipelined) HIP_TRY(hipStreamWaitEvent(s, c->ev_post[k], 0));
    HIP_TRY(hipGetLastError());
    c->last_n = n;
    return RC_OK;
}

static int check_batch_args(rc_ctx *c, const void *frames, uint32_t n, const void *out, const void *rec, const void *md)
{
    if (!c || !frames || !out || !rec || !md) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (n == 0 || n > c->max_batch) return fail(RC_ERR_BAD_ARG, "n must be in 1..max_batch");
    if (!c->thr_set) return fail(RC_ERR_BAD_ARG, "threshold not set (rc_set_threshold / rc_set_dark)");
    return RC_OK;
}

RC_EXPORT int rc_reduce_compress_batch_async(rc_ctx *c, const uint16_t *frames_dev, uint32_t n, uint32_t first_frame_id,
                                             uint8_t *out_dev, uint64_t out_cap, uint64_t *rec_offsets_dev, uint32_t *md_dev)
{
    int r = check_batch_args(c, frames_dev, n, out_dev, rec_offsets_dev, md_dev);
    if (r != RC_OK) return r;
    RC_ON_DEVICE(c->device);
    return enqueue_batch(c, frames_dev, n, first_frame_id, out_dev, out_cap, rec_offsets_dev, md_dev, false);
}

RC_EXPORT int rc_ctx_set_pipelined(rc_ctx *c, int on)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    RC_ON_DEVICE(c->device);
    HIP_TRY(hipStreamSynchronize(c->pstream));  // the second stage changes streams: drain the old one first
    c->pipelined = on != 0;
    c->pstream = (c->pipelined && c->pstream_masked) ? c->pstream_masked : c->pstream_all;
    return RC_OK;
}
RC_EXPORT int rc_ctx_wait_results(rc_ctx *c, void *hip_stream)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    RC_ON_DEVICE(c->device);
    if (c->post_pending[c->last])
        HIP_TRY(hipStreamWaitEvent(hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->stream, c->ev_post[c->last], 0));
    return RC_OK;
}

RC_EXPORT int rc_ctx_sync(rc_ctx *c)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    RC_ON_DEVICE(c->device);
    HIP_TRY(hipMemcpyAsync(&c->h_status[0], c->sc.status, sizeof(rc::BatchStatus), hipMemcpyDeviceToHost, c->pstream));
    HIP_TRY(hipMemcpyAsync(&c->h_status[1], c->d_first_err, sizeof(rc::BatchStatus), hipMemcpyDeviceToHost, c->pstream));
    HIP_TRY(hipMemsetAsync(c->d_first_err, 0, sizeof(rc::BatchStatus), c->pstream));
    HIP_TRY(hipStreamSynchronize(c->pstream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t n_batches = c->batch_seq;
    c->batch_seq = 0;
    for (size_t b = 0; b + 5 <= c->prof_used; b += 5) {  // fold the finished batches' stage events into the sums
        float ms;
        if (!c->profile_all) {
            if (hipEventElapsedTime(&ms, c->prof_ev[b], c->prof_ev[b + 1]) == hipSuccess) c->prof_sum_ms[0] += ms;
        } else
        for (int i = 0; i < 4; ++i)
            if (hipEventElapsedTime(&ms, c->prof_ev[b + i], c->prof_ev[b + i + 1]) == hipSuccess) c->prof_sum_ms[i] += ms;
        if (hipEventElapsedTime(&ms, c->prof_ev[b], c->prof_ev[b + 4]) == hipSuccess) c->prof_sum_ms[4] += ms;
        ++c->prof_batches;
    }
    c->prof_used = 0;
    if (c->h_status[1].code != 0) {  // the first batch that failed since the last sync (not only the most recent one)
        char msg[160];
        if (n_batches > 1)
            snprintf(msg, sizeof msg, "%s (frame %u of batch %llu of the %u enqueued since the last sync)", rc_strerror(c->h_status[1].code),
                     c->h_status[1].frame, (unsigned long long)c->h_status[1].total, n_batches);
        else
            snprintf(msg, sizeof msg, "%s (frame %u of the batch)", rc_strerror(c->h_status[1].code), c->h_status[1].frame);
        return fail(c->h_status[1].code, msg);
    }
    return RC_OK;
}

RC_EXPORT int rc_reduce_compress_batch(rc_ctx *c, const uint16_t *frames, uint32_t n, uint32_t first_frame_id, uint8_t *out,
                                       uint64_t out_cap, uint64_t *rec_offsets, uint32_t *md)
{
    int r = check_batch_args(c, frames, n, out, rec_offsets, md);
    if (r != RC_OK) return r;
    RC_ON_DEVICE(c->device);
    const uint64_t frame_bytes = c->sc.N * 2;
    const uint16_t *fdev = frames;
    if (!is_device_ptr(frames)) {
        r = ensure(c->d_frames, c->d_frames_cap, (uint64_t)n * frame_bytes);
        if (r != RC_OK) return r;
        HIP_TRY(hipMemcpyAsync(c->d_frames, frames, (uint64_t)n * frame_bytes, hipMemcpyHostToDevice, c->stream));
        fdev = c->d_frames;
    }
    uint8_t *odev = out;
    uint64_t cap = out_cap;
    const bool out_host = !is_device_ptr(out);
    if (out_host) {
        const uint64_t worst = (uint64_t)n * frame_bytes;
        cap = out_cap < worst ? out_cap : worst;
        r = ensure(c->d_out, c->d_out_cap, cap);
        if (r != RC_OK) return r;
        odev = c->d_out;
    }
    r = enqueue_batch(c, fdev, n, first_frame_id, odev, cap, c->d_rec_off, c->d_md, true);
    if (r != RC_OK) return r;
    r = rc_ctx_sync(c);
    if (r == RC_ERR_WORKSPACE && c->level == 2) {
        // more foreground than the level-2 workspace holds: the batch's total is known now - grow (with headroom) and run it again
        uint64_t total = 0;
        HIP_TRY(hipMemcpy(&total, c->l2.frame_base + n, 8, hipMemcpyDeviceToHost));
        if (total > c->l2.cap) {
            r = l2_alloc(c, total + total / 4 + 4096);
            if (r != RC_OK) return r;
            r = enqueue_batch(c, fdev, n, first_frame_id, odev, cap, c->d_rec_off, c->d_md, true);
            if (r != RC_OK) return r;
            r = rc_ctx_sync(c);
        }
    }
    for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&c->stage_ms[i], c->ev[i], c->ev[i + 1]);
    (void)hipEventElapsedTime(&c->stage_ms[4], c->ev[0], c->ev[4]);
    if (r != RC_OK) return r;
    int r2 = copy_out(rec_offsets, c->d_rec_off, (uint64_t)(n + 1) * 8, c->stream);
    if (r2 == RC_OK) r2 = copy_out(md, c->d_md, (uint64_t)n * 12, c->stream);
    if (r2 == RC_OK && out_host) r2 = copy_out(out, c->d_out, c->h_status->total, c->stream);
    if (r2 != RC_OK) return r2;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RC_OK;
}

// ---- seam 1, host streaming form ----------------------------------------------------------------------------------------
RC_EXPORT void *rc_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(RC_ERR_DEVICE, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}
RC_EXPORT int rc_host_free(void *p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return RC_OK;
}
RC_EXPORT int rc_host_register(void *p, uint64_t bytes)
{
    if (!p || !bytes) return fail(RC_ERR_BAD_ARG, "NULL / zero argument");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return RC_OK;
}
RC_EXPORT int rc_host_unregister(void *p)
{
    if (p) HIP_TRY(hipHostUnregister(p));
    return RC_OK;
}

static int pipe_slot_init(rc_ctx *c, rc_ctx::PipeSlot &p)
{
    const uint64_t B = c->max_batch;
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->d2h_stream) HIP_TRY(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void **)&p.d_in, B * c->sc.N * 2 + 64));
    HIP_TRY(hipMalloc((void **)&p.d_out, rc_out_capacity(c, (uint32_t)B) + 64));
    HIP_TRY(hipMalloc((void **)&p.d_rec, (B + 1) * 8));
    HIP_TRY(hipMalloc((void **)&p.d_md, B * 12));
    HIP_TRY(hipHostMalloc((void **)&p.h_rec, (B + 1) * 8, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&p.h_md, B * 12, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&p.h_stat, sizeof(rc::BatchStatus), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_h2d, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_fetch, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_val, hipEventDisableTiming));
    HIP_TRY(hipMalloc((void **)&p.d_val, B * 4 + 64));
    HIP_TRY(hipHostMalloc((void **)&p.h_val, B * 4 + 64, hipHostMallocDefault));
    return RC_OK;
}

RC_EXPORT int rc_ctx_set_validation(rc_ctx *c, uint32_t gap, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    if (gap && (w == 0 || h == 0 || w > 128 || h > 128 || (uint64_t)x0 + w > c->nx || (uint64_t)y0 + h > c->ny))
        return fail(RC_ERR_BAD_ARG, "rc_ctx_set_validation: the region must lie inside the frame and hold at most 128 x 128 pixels");
    c->val_gap = gap; c->val_x0 = x0; c->val_y0 = y0; c->val_w = w; c->val_h = h;
    return RC_OK;
}

RC_EXPORT int rc_pipe_validation(rc_ctx *c, uint32_t slot, uint32_t *counts)
{
    if (!c || slot >= RC_PIPE_SLOTS || !counts) return fail(RC_ERR_BAD_ARG, "NULL argument / slot out of range");
    rc_ctx::PipeSlot &p = c->pipe[slot];
    if (p.state != 1 && p.state != 2) return fail(RC_ERR_BAD_ARG, "nothing submitted on this slot");
    if (!p.has_val) { for (uint32_t i = 0; i < p.n; ++i) counts[i] = 0xFFFFFFFFu; return RC_OK; }
    HIP_TRY(hipEventSynchronize(p.ev_val));
    memcpy(counts, p.h_val, (uint64_t)p.n * 4);
    return RC_OK;
}

RC_EXPORT int rc_pipe_submit(rc_ctx *c, uint32_t slot, const uint16_t *frames_host, uint32_t n, uint32_t first_frame_id)
{
    if (!c || !frames_host || slot >= RC_PIPE_SLOTS) return fail(RC_ERR_BAD_ARG, "NULL argument / slot out of range");
    if (n == 0 || n > c->max_batch) return fail(RC_ERR_BAD_ARG, "n must be in 1..max_batch");
    if (!c->thr_set) return fail(RC_ERR_BAD_ARG, "threshold not set (rc_set_threshold / rc_set_dark)");
    RC_ON_DEVICE(c->device);
    rc_ctx::PipeSlot &p = c->pipe[slot];
    if (p.state != 0) return fail(RC_ERR_BAD_ARG, "slot is still in use (result / fetch_wait not called)");
    if (!p.d_in) {
        int r = pipe_slot_init(c, p);
        if (r != RC_OK) return r;
    }
    if (!c->pipelined) {   // the streaming form always lets consecutive batches overlap
        HIP_TRY(hipStreamSynchronize(c->pstream));
        c->pipelined = true;
        c->pstream = c->pstream_masked ? c->pstream_masked : c->pstream_all;
    }
    // Page-locked (or registered) frames are read by the reduce kernel IN PLACE, over the link: every frame byte is needed
    // exactly once, by wide nontemporal loads, so a copy into device memory first would only add a pass (and the copy
    // engines moved 26-31 GB/s here where the kernel's own reads move what the link gives).  RC_PIPE_COPY=1 forces the copy.
    const uint16_t *fdev = nullptr;
    static const bool force_copy = getenv("RC_PIPE_COPY") != nullptr;
    if (!force_copy) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, frames_host) == hipSuccess) {
            if (a.type == hipMemoryTypeHost && a.devicePointer) fdev = reinterpret_cast<const uint16_t *>(a.devicePointer);
        } else (void)hipGetLastError();
    }
    p.zero_copy = fdev != nullptr;
    if (!fdev) {
        HIP_TRY(hipMemcpyAsync(p.d_in, frames_host, (uint64_t)n * c->sc.N * 2, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(p.ev_h2d, c->copy_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, p.ev_h2d, 0));
        fdev = p.d_in;
    }
    int r = enqueue_batch(c, fdev, n, first_frame_id, p.d_out, rc_out_capacity(c, c->max_batch), p.d_rec, p.d_md, false);
    if (r != RC_OK) return r;
    p.has_val = c->val_gap != 0;
    if (p.has_val) {   // validation frames of this batch: the dose-rate count, from the frames the reduce kernel has just read
        rc::launch_roi_components(fdev, c->sc.thr, c->sc.N, c->nx, n, first_frame_id, c->val_gap, c->val_x0, c->val_y0, c->val_w, c->val_h, p.d_val, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(p.h_val, p.d_val, (uint64_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(p.ev_val, c->stream));
    }
    if (p.zero_copy) HIP_TRY(hipEventRecord(p.ev_h2d, c->stream));   // "input consumed" = the reduce kernel (and the count) have run
    hipStream_t ps = c->pstream;   // carries the batch's assembly: the metadata follows it
    HIP_TRY(hipMemcpyAsync(p.h_rec, p.d_rec, (uint64_t)(n + 1) * 8, hipMemcpyDeviceToHost, ps));
    HIP_TRY(hipMemcpyAsync(p.h_md, p.d_md, (uint64_t)n * 12, hipMemcpyDeviceToHost, ps));
    HIP_TRY(hipMemcpyAsync(p.h_stat, c->sc.status, sizeof(rc::BatchStatus), hipMemcpyDeviceToHost, ps));
    HIP_TRY(hipEventRecord(p.ev_done, ps));
    p.n = n;
    p.state = 1;
    return RC_OK;
}

RC_EXPORT int rc_pipe_input_done(rc_ctx *c, uint32_t slot)
{
    if (!c || slot >= RC_PIPE_SLOTS) return fail(RC_ERR_BAD_ARG, "NULL argument / slot out of range");
    if (c->pipe[slot].state == 0) return RC_OK;
    HIP_TRY(hipEventSynchronize(c->pipe[slot].ev_h2d));
    return RC_OK;
}

RC_EXPORT int rc_pipe_result(rc_ctx *c, uint32_t slot, uint64_t *rec_offsets, uint32_t *md, uint64_t *total)
{
    if (!c || slot >= RC_PIPE_SLOTS || !rec_offsets || !md || !total) return fail(RC_ERR_BAD_ARG, "NULL argument / slot out of range");
    rc_ctx::PipeSlot &p = c->pipe[slot];
    if (p.state != 1) return fail(RC_ERR_BAD_ARG, "nothing submitted on this slot");
    HIP_TRY(hipEventSynchronize(p.ev_done));
    p.state = 2;
    if (p.h_stat->code != 0) {
        char msg[128];
        snprintf(msg, sizeof msg, "%s (frame %u of the batch)", rc_strerror(p.h_stat->code), p.h_stat->frame);
        p.state = 0;
        return fail(p.h_stat->code, msg);
    }
    memcpy(rec_offsets, p.h_rec, (uint64_t)(p.n + 1) * 8);
    memcpy(md, p.h_md, (uint64_t)p.n * 12);
    *total = p.h_stat->total;
    return RC_OK;
}

RC_EXPORT int rc_pipe_fetch(rc_ctx *c, uint32_t slot, uint8_t *dst_host, uint64_t bytes)
{
    if (!c || slot >= RC_PIPE_SLOTS || (!dst_host && bytes)) return fail(RC_ERR_BAD_ARG, "NULL argument / slot out of range");
    rc_ctx::PipeSlot &p = c->pipe[slot];
    if (p.state != 2) return fail(RC_ERR_BAD_ARG, "rc_pipe_result has not been called for this slot");
    if (bytes > p.h_stat->total) return fail(RC_ERR_BAD_ARG, "more bytes than the batch's records hold");
    RC_ON_DEVICE(c->device);
    if (bytes) HIP_TRY(hipMemcpyAsync(dst_host, p.d_out, bytes, hipMemcpyDeviceToHost, c->d2h_stream));
    HIP_TRY(hipEventRecord(p.ev_fetch, c->d2h_stream));
    p.state = 3;
    return RC_OK;
}

RC_EXPORT int rc_pipe_fetch_wait(rc_ctx *c, uint32_t slot)
{
    if (!c || slot >= RC_PIPE_SLOTS) return fail(RC_ERR_BAD_ARG, "NULL argument / slot out of range");
    rc_ctx::PipeSlot &p = c->pipe[slot];
    if (p.state == 2) { p.state = 0; return RC_OK; }   // nothing fetched: the slot is simply released
    if (p.state != 3) return fail(RC_ERR_BAD_ARG, "rc_pipe_fetch has not been called for this slot");
    HIP_TRY(hipEventSynchronize(p.ev_fetch));
    p.state = 0;
    return RC_OK;
}

RC_EXPORT int rc_ctx_refit_model(rc_ctx *c)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    c->model_ready = false;
    return RC_OK;
}

RC_EXPORT int rc_get_binary_map(rc_ctx *c, uint32_t i, uint8_t *bitmap_out)
{
    if (!c || !bitmap_out) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (i >= c->last_n) return fail(RC_ERR_BAD_ARG, "frame index outside the most recent batch");
    if (!c->keep_bitmap && c->emit != 0 && c->level != 2) return fail(RC_ERR_BAD_ARG, "binary maps are not kept (rc_ctx_keep_binary_maps(ctx, 0))");
    RC_ON_DEVICE(c->device);
    int r = copy_out(bitmap_out, c->sc.bitmap + (uint64_t)i * c->sc.nb_stride, c->sc.nb, c->stream);
    if (r != RC_OK) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RC_OK;
}

RC_EXPORT int rc_ctx_set_profiling(rc_ctx *c, int on)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    c->profiling = on != 0;
    for (double &v : c->prof_sum_ms) v = 0;
    c->prof_batches = 0;
    return RC_OK;
}
RC_EXPORT int rc_ctx_get_profile(rc_ctx *c, double sum_ms[5], uint64_t *batches)
{
    if (!c || !sum_ms || !batches) return fail(RC_ERR_BAD_ARG, "NULL argument");
    memcpy(sum_ms, c->prof_sum_ms, sizeof c->prof_sum_ms);
    *batches = c->prof_batches;
    return RC_OK;
}

RC_EXPORT int rc_ctx_set_l2_statistics(rc_ctx *c, uint32_t l2_statistics)
{
    if (!c || l2_statistics > 2) return fail(RC_ERR_BAD_ARG, "l2_statistics must be 0, 1 (max) or 2 (sum)");
    c->l2_sum = l2_statistics == 2 ? 1u : 0u;
    return RC_OK;
}

RC_EXPORT int rc_ctx_keep_binary_maps(rc_ctx *c, int on)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    c->keep_bitmap = on != 0;
    return RC_OK;
}

RC_EXPORT int rc_get_stage_ms(rc_ctx *c, float ms[5])
{
    if (!c || !ms) return fail(RC_ERR_BAD_ARG, "NULL argument");
    memcpy(ms, c->stage_ms, sizeof c->stage_ms);
    return RC_OK;
}

// ---- utility contexts for the stateless seams (2 and 3) ----------------------------------------------------------
// One per GPU, created on first use.  A call runs on RC_DEVICE (env) when that is set, otherwise on the CALLER'S CURRENT
// device - in a one-process-per-GPU job that is the rank's own GPU - and leaves the current device as it found it.
namespace {
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
WorkerPool *g_pool = new WorkerPool;

// Growable array in page-locked host memory (a hipMemcpyAsync from it is a real asynchronous copy; capacity is kept).
template <class T>
struct PinnedVec {
    T *p = nullptr; size_t n = 0, cap = 0;
    bool ok = true;                     // false: an allocation failed (checked by the caller after the indexing threads have joined)
    void clear() { n = 0; ok = true; }
    size_t size() const { return n; }
    const T *data() const { return p; }
    void push_back(const T &v)
    {
        if (n == cap) {
            const size_t nc = cap ? cap * 2 : 8192;
            T *q = nullptr;
            if (hipHostMalloc((void **)&q, nc * sizeof(T), hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
            if (n) memcpy(q, p, n * sizeof(T));
            if (p) (void)hipHostFree(p);
            p = q; cap = nc;
        }
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
Util g_utils[RC_MAX_DEV];
thread_local Util *t_util = nullptr;
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

// ---- seam 3 ------------------------------------------------------------------------------------------------------
RC_EXPORT int64_t rc_unpack_frame_sparse(uint32_t nx, uint32_t ny, uint32_t bit_depth, const uint8_t *bitmap,
                                         const uint8_t *pixvals, uint64_t pixvals_bytes, uint64_t *out,
                                         uint64_t out_cap_triplets, uint32_t reduction_level)
{
    if (!bitmap || nx == 0 || ny == 0 || (!out && out_cap_triplets)) return fail(RC_ERR_BAD_ARG, "NULL / zero argument");
    if (reduction_level == 1 && (bit_depth == 0 || bit_depth > 64)) return fail(RC_ERR_BAD_ARG, "bit_depth must be 1..64");
    if (reduction_level == 1 && !pixvals && pixvals_bytes) return fail(RC_ERR_BAD_ARG, "pixvals is NULL");
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    Util &u = g_util;
    const uint64_t N = (uint64_t)nx * ny, nb = (N + 7) / 8, nb8 = (nb + 7) / 8;
    const uint32_t nblk = (uint32_t)((nb8 + rc::WG - 1) / rc::WG);
    const uint8_t *d_bm = nullptr, *d_px = nullptr;
    r = stage_in(bitmap, nb, u.a, u.a_cap, d_bm, nb8 * 8 - nb + 8);
    if (r != RC_OK) return r;
    if (reduction_level == 1 && pixvals_bytes) {
        r = stage_in(pixvals, pixvals_bytes, u.b, u.b_cap, d_px);
        if (r != RC_OK) return r;
    }
    r = ensure(u.w, u.w_cap, (uint64_t)nblk * 8 + 16);
    if (r != RC_OK) return r;
    uint32_t *blk_cnt = reinterpret_cast<uint32_t *>(u.w), *blk_off = blk_cnt + nblk;
    uint64_t *nnz_dev = reinterpret_cast<uint64_t *>(blk_cnt + 2 * (uint64_t)nblk);
    // pass 1: count, so the output can be bounds-checked (and sized) before anything is written
    rc::launch_expand_count(d_bm, nb8, N, blk_cnt, blk_off, nnz_dev, u.stream);
    HIP_TRY(hipMemcpyAsync(u.h_scalar, nnz_dev, 8, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    const uint64_t nnz = *u.h_scalar;
    if (!out) return (int64_t)nnz;  // counting call
    if (nnz > out_cap_triplets) return fail(RC_ERR_OUT_TOO_SMALL, "out holds fewer triplets than the bitmap has set bits");
    if (reduction_level == 1 && (nnz * bit_depth + 7) / 8 > pixvals_bytes)
        return fail(RC_ERR_CORRUPT, "pixvals shorter than popcount(bitmap) * bit_depth bits");
    if (nnz == 0) return 0;
    uint64_t *d_out = out;
    const bool out_host = !is_device_ptr(out);
    if (out_host) {
        r = ensure(u.o, u.o_cap, nnz * 24);
        if (r != RC_OK) return r;
        d_out = reinterpret_cast<uint64_t *>(u.o);
    }
    rc::launch_expand_emit(d_bm, nb8, N, nx, blk_off, d_px, pixvals_bytes, bit_depth, reduction_level, nnz, d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(out, d_out, nnz * 24, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    return (int64_t)nnz;
}

// ---- seam 3, batched: decode + expand n stored frames ---------------------------------------------------------------------
namespace {
// LZ4 frame of independent blocks -> block table (compressed blocks in `comp`, stored ones in `raw`); expect: bytes every
// compressed block regenerates (the last one the rest)
template <class VC, class VR>
int lz4_index_frame(const uint8_t *base, uint64_t off, uint64_t n, uint32_t frame_idx, uint32_t expect, uint64_t total_expected,
                    VC &comp, VR &raw, uint64_t *total)
{
    using namespace rc;
    const uint8_t *p = base + off;
    auto rd32 = [&](uint64_t q) { return (uint32_t)p[q] | ((uint32_t)p[q + 1] << 8) | ((uint32_t)p[q + 2] << 16) | ((uint32_t)p[q + 3] << 24); };
    if (n < 11 || rd32(0) != 0x184D2204u) return ZD_CORRUPT;
    const uint32_t flg = p[4], bd = p[5];
    if ((flg >> 6) != 1 || (flg & 2) || (bd & 0x8F)) return ZD_CORRUPT;
    if (!((flg >> 5) & 1)) return ZD_FOREIGN;                          // linked blocks: a serial chain
    const int bsum = (flg >> 4) & 1, csize = (flg >> 3) & 1, csum = (flg >> 2) & 1, dict = flg & 1;
    uint64_t q = 6 + (csize ? 8 : 0) + (dict ? 4 : 0) + 1, out = 0;
    for (;;) {
        if (q + 4 > n) return ZD_CORRUPT;
        uint32_t bs = rd32(q);
        q += 4;
        if (bs == 0) break;
        const bool stored = bs >> 31;
        bs &= 0x7FFFFFFFu;
        if (q + bs > n) return ZD_CORRUPT;
        // the walk is a chain of dependent cache misses (a header per few lines): ask for the lines a few blocks ahead, assuming
        // blocks of about this size
        if (bs < 2048) { __builtin_prefetch(p + q + 4 * (uint64_t)(bs + 4)); __builtin_prefetch(p + q + 4 * (uint64_t)(bs + 4) + 64); }
        ZdBlock b;
        memset(&b, 0, sizeof b);
        b.frame = frame_idx; b.src = off + q; b.csize = bs; b.dst = (uint32_t)out;
        if (stored) { b.type = 0; b.regen = bs; raw.push_back(b); }
        else {
            if (!expect) return ZD_FOREIGN;
            b.type = 2;
            b.regen = (uint32_t)std::min<uint64_t>(expect, total_expected - out);
            comp.push_back(b);
        }
        out += b.regen;
        if (out > total_expected) return ZD_CORRUPT;
        q += bs + (bsum ? 4 : 0);
    }
    if (csum) q += 4;
    if (q != n) return ZD_CORRUPT;
    *total = out;
    return ZD_OK;
}
}  // namespace

// slot, submit_only: rc_expand_frames = (RC_READ_SLOTS - its own resources -, false); rc_expand_frames_submit = (slot, true): returns once everything is queued.
static int expand_run(uint32_t slot, bool submit_only, uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t level, uint32_t op_mode, uint32_t scheme,
                      const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *nnz_prefix, uint64_t *triplets, uint64_t cap)
{
    using namespace rc;
    if (!data || !sizes || (!nnz_prefix && !submit_only) || n == 0 || nx == 0 || ny == 0 || (!triplets && cap)) return fail(RC_ERR_BAD_ARG, "NULL / zero argument");
    if (level != 1 && level != 3) return fail(RC_ERR_UNSUPPORTED, "rc_expand_frames: reduction level 1 or 3");
    if (level == 1 && (bit_depth == 0 || bit_depth > 64)) return fail(RC_ERR_BAD_ARG, "bit_depth must be 1..64");
    const int codec = op_mode == 0 ? 0 : (scheme == RC_SCHEME_LZ4 ? 2 : (scheme == RC_SCHEME_ZSTD ? 1 : -1));
    if (codec < 0) return fail(RC_ERR_UNSUPPORTED, "rc_expand_frames: scheme has no batched device decoder");
    const uint64_t N = (uint64_t)nx * ny, nb = (N + 7) / 8, nb8 = (nb + 7) / 8;
    static const bool timing = getenv("RC_READ_TIMING") != nullptr;   // development: phase times on stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = now();
    // ---- sizes known without looking at the streams; the copy-in of the compressed bytes starts before the host walks them ----
    const uint64_t bm_stride = nb8 * 8 + 8;
    uint64_t pv_stride = 16, total_in = 0;
    std::vector<uint64_t> foff(n);
    for (uint32_t f = 0; f < n; ++f) {
        const uint32_t npk = level == 1 ? sizes[3 * f + 2] : 0;
        pv_stride = std::max<uint64_t>(pv_stride, ((uint64_t)npk + 15) & ~15ull);
        foff[f] = total_in;
        total_in += (uint64_t)sizes[3 * f] + (level == 1 ? sizes[3 * f + 1] : 0);
    }
    pv_stride += 16;
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    Util &U = g_util;
    ReadRes &u = U.rr[slot];
    if (u.pending) return fail(RC_ERR_BAD_ARG, "rc_expand_frames: this slot holds a submitted batch - rc_expand_frames_wait first");
    if (!u.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&u.stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&u.stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&u.ev_a, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&u.ev_b, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&u.done, hipEventDisableTiming));
    }
    if (u.h_res_cap < (uint64_t)n + 2) {
        if (u.h_res) HIP_TRY(hipHostFree(u.h_res));
        u.h_res = nullptr; u.h_res_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&u.h_res, ((uint64_t)n + 2) * 8, hipHostMallocDefault));
        u.h_res_cap = (uint64_t)n + 2;
    }
    hipStream_t s = u.stream;
    const uint32_t nblk = (uint32_t)((nb8 + WG - 1) / WG);
    const uint64_t out_bytes = (uint64_t)n * (bm_stride + (level == 1 ? pv_stride : 0)) + 64;
    auto need = [&](int i, uint64_t bytes) { return ensure(u.x[i], u.x_cap[i], bytes); };
    // head: [ZdTables bitmap x n][ZdTables values x n] (zstd) [block lists: bitmap x n, values x n, stored x threads, compact bitmap x n][pv_bytes n] [base2 2n][pv_base n][src_base n], the
    // same layout in page-locked host memory and on the device: one copy
    const uint64_t ntab = codec == 1 ? 2 * (uint64_t)n : 0;
    const uint64_t o_first = ntab * sizeof(ZdTables);
    const uint64_t o_base2 = (o_first + (3 * (uint64_t)n + RC_READ_THREADS) * sizeof(ZdFrameList) + (uint64_t)n * 4 + 15) & ~15ull;
    const uint64_t sz_head = o_base2 + (uint64_t)n * 4 * 8;
    if ((r = need(0, total_in + 64)) != RC_OK || (r = need(1, out_bytes)) != RC_OK || (r = need(3, sz_head)) != RC_OK ||
        (r = need(5, (uint64_t)n * nblk * 8 + (uint64_t)(2 * n + 2) * 8 + 64)) != RC_OK)
        return r;
    if (u.rd_head_cap < sz_head) {
        if (u.rd_head) HIP_TRY(hipHostFree(u.rd_head));
        u.rd_head = nullptr; u.rd_head_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&u.rd_head, sz_head, hipHostMallocDefault));
        u.rd_head_cap = sz_head;
    }
    if (!U.zd_predef) {
        std::vector<uint8_t> t(zd_tables_bytes());
        zd_predefined_tables(t.data());
        HIP_TRY(hipMalloc(&U.zd_predef, t.size()));
        HIP_TRY(hipMemcpy(U.zd_predef, t.data(), t.size(), hipMemcpyHostToDevice));
    }
    // The compressed bytes: device memory is used where it lies; host memory is copied in, and the copy runs while the host walks the
    // streams.  (Letting the decoders read page-locked host memory in place - their staging loads as the transfer - was slower: the
    // transfer then sits inside the decoders' critical path, 1.3 ms against 0.7 ms behind a copy that hides under the host walk.)
    const uint8_t *d_data = u.x[0];
    uint8_t *d_out = u.x[1];
    bool copy_in = true;
    {
        // in place only if the decoders' 16-byte staging loads (and the bit readers' dword loads) can neither be misaligned nor leave
        // the last page of the caller's allocation: they may touch up to 15 bytes behind the last stream
        const uintptr_t end = (uintptr_t)data + total_in;
        const bool usable = ((uintptr_t)data & 15u) == 0 && (end & 4095u) != 0 && (end & 4095u) <= 4096u - 16u;
        if (usable && is_device_ptr(data)) { d_data = data; copy_in = false; }
    }
    uint32_t *d_blk_cnt = reinterpret_cast<uint32_t *>(u.x[5]), *d_blk_off = d_blk_cnt + (uint64_t)n * nblk;
    uint64_t *d_fnnz = reinterpret_cast<uint64_t *>(d_blk_off + (uint64_t)n * nblk), *d_fbase = d_fnnz + n;
    int *d_err = reinterpret_cast<int *>(d_fbase + n + 1);
    // The header walk below runs on the host.  Bytes that lie in device memory are fetched once into page-locked memory for it (the
    // host CAN read device memory through the PCIe aperture, a few hundred MB/s: 187 ms for 34 MB); the decoders read them where they are.
    const uint8_t *walk = data;
    if (is_device_ptr(data)) {
        if (u.h_blob_cap < total_in + 64) {
            if (u.h_blob) HIP_TRY(hipHostFree(u.h_blob));
            u.h_blob = nullptr; u.h_blob_cap = 0;
            HIP_TRY(hipHostMalloc((void **)&u.h_blob, total_in + 64 + total_in / 4, hipHostMallocDefault));
            u.h_blob_cap = total_in + 64 + total_in / 4;
        }
        HIP_TRY(hipMemcpyAsync(u.h_blob, data, total_in, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        walk = u.h_blob;
    }
    if (copy_in) HIP_TRY(hipMemcpyAsync(u.x[0], data, total_in, hipMemcpyDefault, s));
    HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, s));   // bitmap padding and value-stream tails read as zero
    HIP_TRY(hipMemsetAsync(d_err, 0, 4, s));
    // ---- host: walk the frames, build block tables and decoding tables (a few threads, each a contiguous range of frames) ----
    // (the copy-in reads the caller's memory: no return from here on without waiting for it)
    auto bail = [&](int code, const char *msg) { (void)hipStreamSynchronize(s); return fail(code, msg); };
    ZdTables *bm_tab = reinterpret_cast<ZdTables *>(u.rd_head), *pv_tab = bm_tab + (codec == 1 ? n : 0);
    // The block lists stay where the indexing threads wrote them, in page-locked host memory: the decoders read every entry once,
    // over the link (uploading them meant 3 small copies per thread, each a fixed ~15 us of stream time: 0.7 ms per call).
    ZdFrameList *bm_list = reinterpret_cast<ZdFrameList *>(u.rd_head + o_first), *pv_list = bm_list + n, *raw_list = pv_list + n;
    ZdFrameList *cbm_list = raw_list + RC_READ_THREADS;
    uint32_t *pv_bytes = reinterpret_cast<uint32_t *>(cbm_list + n);
    uint64_t *base2 = reinterpret_cast<uint64_t *>(u.rd_head + o_base2), *pv_base = base2 + 2 * (uint64_t)n, *src_base = pv_base + n;
    // c0, c_n: the frame's range in its thread's compact offset list (c_n blocks = c_n + 1 offsets); c_skips: tree_skip | seq_skip << 8
    struct FrameIndex { uint32_t bm0 = 0, bm_n = 0, pv0 = 0, pv_n = 0, thread = 0, c0 = 0, c_n = 0, c_skips = 0; int status = ZD_OK; const char *what = nullptr; };
    std::vector<FrameIndex> fi(n);
    const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    static const uint32_t thr_env = getenv("RC_READ_THREADS") ? (uint32_t)atoi(getenv("RC_READ_THREADS")) : 0u;   // (development: 1..16)
    const uint32_t nthr = std::max(1u, std::min<uint32_t>(std::min<uint32_t>(n, thr_env ? std::min<uint32_t>(thr_env, RC_READ_THREADS) : RC_READ_THREADS), hw));
    const int dev_now = U.device;
    // frames are claimed one at a time: the calling thread starts at once, the pool's workers join in as they wake up (their
    // wake-up, not the walk - 30 us per frame - is what a static split waited for)
    std::atomic<uint32_t> next_frame{0};
    auto index_range = [&](uint32_t t) {
        if (t) (void)hipSetDevice(dev_now);   // (a worker thread: page-locked memory it allocates belongs to this device's context)
        auto &BM = u.rd_bm[t]; auto &PV = u.rd_pv[t]; auto &RAW = u.rd_raw[t]; auto &all = u.rd_tmp[t]; auto &OFF = u.rd_off[t];
        BM.clear(); PV.clear(); RAW.clear(); OFF.clear();
        // A binary-map stream whose blocks all regenerate TILE_BM bytes (the last one the rest), lie back to back and keep to one
        // set of sequence tables - what this library's encoders write - leaves one dword per block (k_bitmap_decode_c); any other
        // stream inside the decoders' subset leaves full entries, Compressed blocks and stored ones apart, as before.
        auto route_bitmap = [&](FrameIndex &F, uint64_t o, uint64_t cb, uint32_t hdr, bool one_table_set) {
            bool uniform = one_table_set && !all.empty() && cb < (1ull << 32);
            uint32_t skips = 0;
            for (size_t i = 0; uniform && i < all.size(); ++i) {
                const ZdBlock &b = all[i];
                const uint64_t want = std::min<uint64_t>((uint64_t)TILE_BM, nb - std::min<uint64_t>(nb, (uint64_t)i * TILE_BM));
                uniform = b.regen == want && b.dst == (uint64_t)i * TILE_BM && (i + 1 == all.size() || all[i + 1].src - hdr == b.src + b.csize);
                if (b.tree_skip) skips |= b.tree_skip;
                if (b.seq_skip > 1) skips |= (uint32_t)b.seq_skip << 8;     // (1 = the RLE offset byte of a block with predefined tables)
            }
            if (uniform) {
                F.c0 = (uint32_t)OFF.size(); F.c_n = (uint32_t)all.size(); F.c_skips = skips;
                for (const ZdBlock &b : all) OFF.push_back((uint32_t)(b.src - hdr - o));
                OFF.push_back((uint32_t)(all.back().src + all.back().csize - o));
            } else
                for (const ZdBlock &b : all) { if (b.type == 2) BM.push_back(b); else RAW.push_back(b); }
        };
        for (;;) {
            const uint32_t f = next_frame.fetch_add(1, std::memory_order_relaxed);
            if (f >= n) break;
            FrameIndex &F = fi[f];
            F.thread = t;
            const uint64_t cb = sizes[3 * f], cp = level == 1 ? sizes[3 * f + 1] : 0, npk = level == 1 ? sizes[3 * f + 2] : 0;
            const uint64_t o = foff[f];
            uint64_t got = 0;
            int rr = ZD_OK;
            F.bm0 = (uint32_t)BM.size(); F.pv0 = (uint32_t)PV.size();
            if (codec == 0) {
                if (cb != nb || cp != npk) { F.status = ZD_CORRUPT; F.what = "rc_expand_frames: mode-0 sizes disagree with the frame shape"; continue; }
                ZdBlock b;
                memset(&b, 0, sizeof b);
                b.frame = f; b.src = o; b.csize = b.regen = (uint32_t)nb; b.dst = 0;
                RAW.push_back(b);
                if (npk) { b.src = o + cb; b.csize = b.regen = (uint32_t)npk; b.frame = n + f; RAW.push_back(b); }
            } else if (codec == 2) {
                all.clear();
                rr = lz4_index_frame(walk, o, cb, f, TILE_BM, nb, all, all, &got);
                if (rr == ZD_OK && got != nb) rr = ZD_CORRUPT;
                if (rr == ZD_OK) route_bitmap(F, o, cb, 4, true);
                if (rr == ZD_OK && level == 1) {
                    all.clear();   // (a value stream holds stored chunks only: a compressed block there is outside the subset)
                    rr = lz4_index_frame(walk, o + cb, cp, n + f, 0, npk, all, RAW, &got);
                    if (rr == ZD_OK && got != npk) rr = ZD_CORRUPT;
                }
            } else {
                // (Compressed blocks to the stream's list, stored / RLE ones to the copy list, as the walk finds them)
                struct Route {
                    PinnedVec<ZdBlock> &comp, &raw;
                    bool values; uint32_t frame; bool too_long = false;
                    void push_back(const ZdBlock &b)
                    {
                        if (b.type != 2) { raw.push_back(b); return; }
                        if (!values) { comp.push_back(b); return; }
                        if (b.regen > 1024) { too_long = true; return; }   // a value-stream block the chunk decoder is not built for
                        ZdBlock c = b;
                        c.frame = frame;
                        comp.push_back(c);
                    }
                };
                Route rp{PV, RAW, true, f};
                all.clear();
                rr = zd_index_frame(walk, o, cb, f, TILE_BM, nb, all, bm_tab[f], &got);
                if (rr == ZD_OK && got != nb) rr = ZD_CORRUPT;
                if (rr == ZD_OK) route_bitmap(F, o, cb, 3, !(bm_tab[f].has & 4u));
                if (rr == ZD_OK && level == 1) {
                    rr = zd_index_frame(walk, o + cb, cp, n + f, 0, npk, rp, pv_tab[f], &got);
                    if (rr == ZD_OK && got != npk) rr = ZD_CORRUPT;
                    if (rr == ZD_OK && rp.too_long) rr = ZD_FOREIGN;
                }
            }
            F.bm_n = (uint32_t)BM.size() - F.bm0; F.pv_n = (uint32_t)PV.size() - F.pv0;
            F.status = rr;
        }
    };
    g_pool->run(nthr, index_range);
    const double t_1 = now();
    uint64_t n_bm = 0, n_pv = 0, n_raw = 0;
    uint32_t bm_max = 0, pv_max = 0, raw_max_regen = 0, cbm_max = 0;
    for (uint32_t t = 0; t < nthr; ++t) {
        if (!u.rd_bm[t].ok || !u.rd_pv[t].ok || !u.rd_raw[t].ok || !u.rd_off[t].ok) return bail(RC_ERR_DEVICE, "rc_expand_frames: page-locked host memory exhausted");
        raw_list[t].p = u.rd_raw[t].data(); raw_list[t].n = (uint32_t)u.rd_raw[t].size(); raw_list[t].pad = 0;
        n_raw += u.rd_raw[t].size();
        const ZdBlock *rb = u.rd_raw[t].data();
        for (size_t i = 0; i < u.rd_raw[t].size(); ++i) raw_max_regen = std::max(raw_max_regen, rb[i].regen);
    }
    for (uint32_t f = 0; f < n; ++f) {
        const FrameIndex &F = fi[f];
        const uint32_t t = F.thread;
        if (F.status == ZD_FOREIGN) return bail(RC_ERR_UNSUPPORTED, "rc_expand_frames: stream outside the device decoders' subset (use the stock decoder)");
        if (F.status != ZD_OK) return bail(RC_ERR_CORRUPT, F.what ? F.what : "rc_expand_frames: malformed compressed stream");
        bm_list[f].p = u.rd_bm[t].data() + F.bm0; bm_list[f].n = F.bm_n; bm_list[f].pad = 0;
        pv_list[f].p = u.rd_pv[t].data() + F.pv0; pv_list[f].n = F.pv_n; pv_list[f].pad = 0;
        cbm_list[f].p = reinterpret_cast<const ZdBlock *>(u.rd_off[t].data() + F.c0); cbm_list[f].n = F.c_n; cbm_list[f].pad = F.c_skips;
        src_base[f] = foff[f];
        cbm_max = std::max(cbm_max, F.c_n);
        pv_bytes[f] = level == 1 ? sizes[3 * f + 2] : 0;
        base2[f] = (uint64_t)f * bm_stride;                                        // stored blocks: frames 0..n-1 = bitmaps,
        base2[n + f] = pv_base[f] = (uint64_t)n * bm_stride + (uint64_t)f * pv_stride;   // n..2n-1 = value streams (behind the bitmaps)
        bm_max = std::max(bm_max, F.bm_n);
        pv_max = std::max(pv_max, F.pv_n);
        n_bm += F.bm_n; n_pv += F.pv_n;
    }
    if (n_raw >= (1ull << 31)) return bail(RC_ERR_UNSUPPORTED, "rc_expand_frames: too many blocks in one call");
    const double t_2 = now();
    // ---- device ----
    ZdTables *d_bm_tab = reinterpret_cast<ZdTables *>(u.x[3]), *d_pv_tab = d_bm_tab + (codec == 1 ? n : 0);
    const ZdFrameList *d_bm_list = reinterpret_cast<const ZdFrameList *>(u.x[3] + o_first), *d_pv_list = d_bm_list + n, *d_raw_list = d_pv_list + n;
    const ZdFrameList *d_cbm_list = d_raw_list + RC_READ_THREADS;
    uint32_t *d_pv_bytes = reinterpret_cast<uint32_t *>(u.x[3] + o_first + (3 * (uint64_t)n + RC_READ_THREADS) * sizeof(ZdFrameList));
    uint64_t *d_base2 = reinterpret_cast<uint64_t *>(u.x[3] + o_base2), *d_pvbase = d_base2 + 2 * (uint64_t)n, *d_src_base = d_pvbase + n;
    HIP_TRY(hipMemcpyAsync(u.x[3], u.rd_head, sz_head, hipMemcpyHostToDevice, s));
    const double t_3 = now();
    // the value streams' chunks (few, long serial chains) decode next to the binary maps' blocks (many, short), on a second stream
    static const bool serial = getenv("RC_READ_SERIAL") != nullptr;   // development: both decoders on one stream (clean per-kernel times)
    if (n_pv && serial) launch_block_decode(1, 1024, d_data, d_pv_list, n, pv_max, d_pv_tab, U.zd_predef, d_out, d_pvbase, d_err, s);
    else if (n_pv) {
        HIP_TRY(hipEventRecord(u.ev_a, s));
        HIP_TRY(hipStreamWaitEvent(u.stream2, u.ev_a, 0));
        launch_block_decode(1, 1024, d_data, d_pv_list, n, pv_max, d_pv_tab, U.zd_predef, d_out, d_pvbase, d_err, u.stream2);
        HIP_TRY(hipEventRecord(u.ev_b, u.stream2));
    }
    if (cbm_max) launch_bitmap_decode_compact(codec == 1 ? 1 : 2, d_data, d_cbm_list, d_src_base, n, cbm_max, d_bm_tab, U.zd_predef, d_out, d_base2, nb, d_err, s);
    if (n_bm) launch_block_decode(codec == 1 ? 1 : 2, TILE_BM, d_data, d_bm_list, n, bm_max, d_bm_tab, U.zd_predef, d_out, d_base2, d_err, s);
    launch_block_copy(d_data, d_raw_list, nthr, (uint32_t)n_raw, raw_max_regen, d_out, d_base2, s);
    if (n_pv && !serial) HIP_TRY(hipStreamWaitEvent(s, u.ev_b, 0));
    const uint8_t *d_bm = d_out, *d_pv = d_out + (uint64_t)n * bm_stride;
    // Triplets wanted in DEVICE memory: the emit kernel is queued right behind the count - no host round trip in between; the kernel that
    // finishes the count (k_expand_bases) checks what the host otherwise would (total <= cap, value streams long enough) and the emit
    // kernel writes nothing once any check or decoder has raised *d_err.  Host memory: the output is staged, so its size must be known
    // first (one more synchronisation).
    // A submitted batch may also name PAGE-LOCKED host memory: the triplets are then staged in device memory and one asynchronous copy of
    // cap entries follows the emit kernel (the copy engine moves 64 MB in 1.3 ms under the next batch's work; letting the emit kernel
    // write over the link itself - 8-byte stores, 24 bytes apart - took 4 ms).
    bool dev_out = triplets && is_device_ptr(triplets);
    uint64_t *host_async = nullptr;
    if (submit_only && !dev_out) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, triplets) == hipSuccess && a.type == hipMemoryTypeHost) {
            if ((r = need(6, cap * 24 + 64)) != RC_OK) { (void)hipStreamSynchronize(s); return r; }
            host_async = triplets;
            triplets = reinterpret_cast<uint64_t *>(u.x[6]);
            dev_out = true;
        } else (void)hipGetLastError();
    }
    if (submit_only && !dev_out) return bail(RC_ERR_BAD_ARG, "rc_expand_frames_submit: triplets must be device or page-locked host memory");
    if (dev_out) {
        launch_expand_batch_count(d_bm, bm_stride, nb8, N, n, d_blk_cnt, d_blk_off, d_fnnz, d_fbase, s, d_pv_bytes, bit_depth, level, cap, d_err);
        launch_expand_batch_emit(d_bm, bm_stride, nb8, N, nx, n, d_blk_off, d_fbase, d_pv, pv_stride, d_pv_bytes, bit_depth, level, cap, triplets, s, d_err);
    } else
        launch_expand_batch_count(d_bm, bm_stride, nb8, N, n, d_blk_cnt, d_blk_off, d_fnnz, d_fbase, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(u.h_res, d_fbase, (uint64_t)(n + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(u.h_res + n + 1, d_err, 4, hipMemcpyDeviceToHost, s));
    if (host_async && cap) HIP_TRY(hipMemcpyAsync(host_async, triplets, cap * 24, hipMemcpyDeviceToHost, s));
    if (submit_only) {   // (dev_out is a precondition, checked above)
        HIP_TRY(hipEventRecord(u.done, s));
        u.pending = true;
        u.n = n; u.level = level; u.bit_depth = bit_depth; u.cap = cap;
        u.pv_bytes.assign(pv_bytes, pv_bytes + n);
        return RC_OK;
    }
    HIP_TRY(hipStreamSynchronize(s));
    const int err = (int)(uint32_t)u.h_res[n + 1];
    memcpy(nnz_prefix, u.h_res, (size_t)(n + 1) * 8);
    const double t_4 = now();
    if (err & 1) return fail(RC_ERR_CORRUPT, "rc_expand_frames: a block does not decode to its expected size");
    const uint64_t total = nnz_prefix[n];
    if (!triplets) return RC_OK;
    if (total > cap || (err & 2)) return fail(RC_ERR_OUT_TOO_SMALL, "rc_expand_frames: triplets holds fewer entries than the frames have set pixels");
    if (level == 1)
        for (uint32_t f = 0; f < n; ++f)
            if (((nnz_prefix[f + 1] - nnz_prefix[f]) * bit_depth + 7) / 8 > pv_bytes[f])
                return fail(RC_ERR_CORRUPT, "rc_expand_frames: value stream shorter than popcount(bitmap) * bit_depth bits");
    if (dev_out) {
        if (timing)
            fprintf(stderr, "rc_expand_frames: index %.3f ms, merge %.3f, enqueue copies %.3f, decode+count+emit (to sync) %.3f\n", t_1 - t_0, t_2 - t_1,
                    t_3 - t_2, t_4 - t_3);
        return RC_OK;
    }
    if (total == 0) return RC_OK;
    if ((r = need(6, total * 24)) != RC_OK) return r;
    uint64_t *d_trip = reinterpret_cast<uint64_t *>(u.x[6]);
    launch_expand_batch_emit(d_bm, bm_stride, nb8, N, nx, n, d_blk_off, d_fbase, d_pv, pv_stride, d_pv_bytes, bit_depth, level, total, d_trip, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(triplets, d_trip, total * 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (timing)
        fprintf(stderr, "rc_expand_frames: index %.3f ms, merge %.3f, enqueue copies %.3f, decode+count (to sync) %.3f, emit %.3f\n", t_1 - t_0, t_2 - t_1,
                t_3 - t_2, t_4 - t_3, now() - t_4);
    return RC_OK;
}

RC_EXPORT int rc_expand_frames(uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t level, uint32_t op_mode, uint32_t scheme,
                               const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *nnz_prefix, uint64_t *triplets, uint64_t cap)
{
    return expand_run(RC_READ_SLOTS, false, nx, ny, bit_depth, level, op_mode, scheme, data, sizes, n, nnz_prefix, triplets, cap);
}

RC_EXPORT int rc_expand_frames_submit(uint32_t slot, uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t level, uint32_t op_mode,
                                      uint32_t scheme, const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *triplets_dev, uint64_t cap)
{
    if (slot >= RC_READ_SLOTS) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_submit: slot 0 or 1");
    if (!triplets_dev) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_submit: triplets must be device or page-locked host memory");
    return expand_run(slot, true, nx, ny, bit_depth, level, op_mode, scheme, data, sizes, n, nullptr, triplets_dev, cap);
}

RC_EXPORT int rc_expand_frames_wait(uint32_t slot, uint64_t *nnz_prefix)
{
    if (slot >= RC_READ_SLOTS || !nnz_prefix) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_wait: slot 0 or 1, nnz_prefix");
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    ReadRes &u = g_util.rr[slot];
    if (!u.pending) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_wait: nothing was submitted to this slot");
    u.pending = false;
    HIP_TRY(hipEventSynchronize(u.done));
    const uint32_t n = u.n;
    const int err = (int)(uint32_t)u.h_res[n + 1];
    memcpy(nnz_prefix, u.h_res, (size_t)(n + 1) * 8);
    if (err & 1) return fail(RC_ERR_CORRUPT, "rc_expand_frames: a block does not decode to its expected size");
    if (nnz_prefix[n] > u.cap || (err & 2)) return fail(RC_ERR_OUT_TOO_SMALL, "rc_expand_frames: triplets holds fewer entries than the frames have set pixels");
    if (err & 4) return fail(RC_ERR_CORRUPT, "rc_expand_frames: value stream shorter than popcount(bitmap) * bit_depth bits");
    return RC_OK;
}

RC_EXPORT int rc_bit_pack(const uint16_t *pixvals, uint64_t n, uint32_t bit_depth, uint8_t *out, uint64_t out_n)
{
    if (!out || (!pixvals && n)) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (bit_depth == 0 || bit_depth > 32) return fail(RC_ERR_BAD_ARG, "bit_depth must be 1..32");
    if (out_n != (n * bit_depth + 7) / 8) return fail(RC_ERR_BAD_ARG, "out_n must be ceil(n*bit_depth/8)");
    if (out_n == 0) return RC_OK;
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    Util &u = g_util;
    const uint16_t *d_in = nullptr;
    r = stage_in(pixvals, n * 2, u.a, u.a_cap, d_in);
    if (r != RC_OK) return r;
    uint8_t *d_out = out;
    const bool out_host = !is_device_ptr(out);
    if (out_host) {
        r = ensure(u.o, u.o_cap, out_n);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    rc::launch_bit_pack(d_in, n, bit_depth, d_out, out_n, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(out, d_out, out_n, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    return RC_OK;
}

RC_EXPORT int rc_bit_unpack(const uint8_t *packed, uint64_t packed_bytes, uint64_t n, uint32_t bit_depth, uint64_t *out)
{
    if (!out || (!packed && packed_bytes)) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (bit_depth == 0 || bit_depth > 64) return fail(RC_ERR_BAD_ARG, "bit_depth must be 1..64");
    if ((n * bit_depth + 7) / 8 > packed_bytes) return fail(RC_ERR_CORRUPT, "packed shorter than n * bit_depth bits");
    if (n == 0) return RC_OK;
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    Util &u = g_util;
    const uint8_t *d_in = nullptr;
    r = stage_in(packed, packed_bytes, u.a, u.a_cap, d_in);
    if (r != RC_OK) return r;
    uint64_t *d_out = out;
    const bool out_host = !is_device_ptr(out);
    if (out_host) {
        r = ensure(u.o, u.o_cap, n * 8);
        if (r != RC_OK) return r;
        d_out = reinterpret_cast<uint64_t *>(u.o);
    }
    rc::launch_bit_unpack(d_in, packed_bytes, n, bit_depth, d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    return RC_OK;
}

// ---- seam 2 ----------------------------------------------------------------------------------------------------
static int lz4_compress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n, uint32_t level)
{
    using namespace rc;
    Util &u = g_util;
    if (n >= (1ull << 32)) return fail(RC_ERR_BAD_ARG, "rc_compress: input must be < 4 GiB");
    if (n == 0) {  // empty frame: header + EndMark
        const uint32_t h = lz4f_descriptor(0x40);
        const uint8_t f[11] = {0x04, 0x22, 0x4D, 0x18, (uint8_t)h, (uint8_t)(h >> 8), (uint8_t)(h >> 16), 0, 0, 0, 0};
        if (dst_cap < 11) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small");
        HIP_TRY(hipMemcpy(dst, f, 11, is_device_ptr(dst) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        *out_n = 11;
        return RC_OK;
    }
    Scratch sc;
    sc.ntiles = (uint32_t)((n + TILE_BM - 1) / TILE_BM);
    sc.nb = n;
    sc.nb_stride = (uint64_t)sc.ntiles * TILE_BM;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src, sc.nb_stride - n + 16);
    if (r != RC_OK) return r;
    sc.bitmap = const_cast<uint8_t *>(d_src);
    const uint64_t T = sc.ntiles;
    r = ensure(u.w, u.w_cap, T * BLK_SLOT + T * 8 + 64);
    if (r != RC_OK) return r;
    sc.blk_slots = u.w;
    sc.blk_size = reinterpret_cast<uint32_t *>(u.w + T * BLK_SLOT);
    sc.blk_off = sc.blk_size + T;
    sc.frame_cbytes = sc.blk_off + T;
    launch_lz4_encode_buffer(sc, u.stream, level != 0);   // level 0: zero runs only; >= 1: the event parser (rc_lz4_block.h)
    launch_scans(sc, 1, false, true, u.stream);
    HIP_TRY(hipMemcpyAsync(u.h_scalar, sc.frame_cbytes, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    const uint64_t total = 7ull + *reinterpret_cast<uint32_t *>(u.h_scalar) + 4;
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small (see rc_compress_bound)");
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_lz4f_gather(sc, lz4f_descriptor(0x40), d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    *out_n = total;
    return RC_OK;
}

static int lz4_decompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    // the frame and block headers are walked on the host (sequential by format, a few bytes per block)
    std::vector<uint8_t> hsrc;
    const uint8_t *h = src;
    if (is_device_ptr(src)) {
        hsrc.resize(n);
        HIP_TRY(hipMemcpy(hsrc.data(), src, n, hipMemcpyDeviceToHost));
        h = hsrc.data();
    }
    auto rd32 = [&](uint64_t p) { return (uint32_t)h[p] | ((uint32_t)h[p + 1] << 8) | ((uint32_t)h[p + 2] << 16) | ((uint32_t)h[p + 3] << 24); };
    if (n < 11 || rd32(0) != 0x184D2204u) return fail(RC_ERR_CORRUPT, "not an LZ4 frame");
    const uint32_t flg = h[4], bd = h[5];
    if ((flg >> 6) != 1 || (flg & 2) || (bd & 0x8F) || ((bd >> 4) & 7) < 4) return fail(RC_ERR_CORRUPT, "bad LZ4 frame descriptor");
    const int linked = !((flg >> 5) & 1), bsum = (flg >> 4) & 1, csize = (flg >> 3) & 1, csum = (flg >> 2) & 1, dict = flg & 1;
    const uint64_t bmax = 1ull << (8 + 2 * ((bd >> 4) & 7));
    uint64_t ip = 6 + (csize ? 8 : 0) + (dict ? 4 : 0) + 1;
    std::vector<Lz4Block> blks;
    for (;;) {
        if (ip + 4 > n) return fail(RC_ERR_CORRUPT, "truncated LZ4 frame");
        uint32_t bs = rd32(ip);
        ip += 4;
        if (bs == 0) break;
        const uint32_t raw = bs >> 31;
        bs &= 0x7FFFFFFFu;
        if (bs > bmax || ip + bs > n) return fail(RC_ERR_CORRUPT, "LZ4 block exceeds the frame");
        blks.push_back(Lz4Block{ip, bs, raw});
        ip += bs + (bsum ? 4 : 0);
    }
    if (csum) ip += 4;
    if (ip != n) return fail(RC_ERR_CORRUPT, "trailing bytes after the LZ4 frame");
    const uint32_t nblk = (uint32_t)blks.size();
    if (nblk == 0) { *out_n = 0; return RC_OK; }
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src);
    if (r != RC_OK) return r;
    // work buffer: block table | sizes | offsets | err
    const uint64_t tab = (uint64_t)nblk * sizeof(Lz4Block), szs = ((uint64_t)nblk * 4 + 7) & ~7ull, offs = (uint64_t)nblk * 8;
    r = ensure(u.w, u.w_cap, tab + szs + offs + 16);
    if (r != RC_OK) return r;
    Lz4Block *d_blks = reinterpret_cast<Lz4Block *>(u.w);
    uint32_t *d_sizes = reinterpret_cast<uint32_t *>(u.w + tab);
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(u.w + tab + szs);
    int *d_err = reinterpret_cast<int *>(u.w + tab + szs + offs);
    HIP_TRY(hipMemcpyAsync(d_blks, blks.data(), tab, hipMemcpyHostToDevice, u.stream));
    HIP_TRY(hipMemsetAsync(d_err, 0, 4, u.stream));
    launch_lz4_decode(d_src, d_blks, nblk, d_sizes, nullptr, nullptr, ~0ull, linked, d_err, u.stream);
    std::vector<uint32_t> sizes(nblk);
    int err = 0;
    HIP_TRY(hipMemcpyAsync(sizes.data(), d_sizes, (uint64_t)(linked ? 1 : nblk) * 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block");
    std::vector<uint64_t> off(nblk, 0);
    uint64_t total = 0;
    if (linked) total = sizes[0];
    else
        for (uint32_t b = 0; b < nblk; ++b) {
            if (sizes[b] > bmax) return fail(RC_ERR_CORRUPT, "LZ4 block decodes beyond its declared maximum");
            off[b] = total;
            total += sizes[b];
        }
    *out_n = total;  // reported even when dst is too small, so a caller can size its buffer and call again
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small");
    if (total == 0) return RC_OK;
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    HIP_TRY(hipMemcpyAsync(d_offs, off.data(), offs, hipMemcpyHostToDevice, u.stream));
    uint32_t max_stored = 0;
    for (const Lz4Block &q : blks) if (q.raw) max_stored = std::max(max_stored, q.size);
    launch_lz4_decode(d_src, d_blks, nblk, nullptr, d_offs, d_out, total, linked, d_err, u.stream, max_stored);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block");
    return RC_OK;
}

static int zstd_compress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    if (n >= (1ull << 32)) return fail(RC_ERR_BAD_ARG, "rc_compress: input must be < 4 GiB");
    if (n == 0) {  // a frame needs one block: empty raw block with Last_Block
        const uint8_t f[9] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x00, 0x01, 0x00, 0x00};
        if (dst_cap < 9) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small");
        HIP_TRY(hipMemcpy(dst, f, 9, is_device_ptr(dst) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        *out_n = 9;
        return RC_OK;
    }
    Scratch sc;
    sc.ntiles = (uint32_t)((n + TILE_BM - 1) / TILE_BM);
    sc.nb = n;
    sc.nb_stride = (uint64_t)sc.ntiles * TILE_BM;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src, sc.nb_stride - n + 16);
    if (r != RC_OK) return r;
    sc.bitmap = const_cast<uint8_t *>(d_src);
    const uint64_t T = sc.ntiles;
    r = ensure(u.w, u.w_cap, T * BLK_SLOT + T * 8 + 64);
    if (r != RC_OK) return r;
    sc.blk_slots = u.w;
    sc.blk_size = reinterpret_cast<uint32_t *>(u.w + T * BLK_SLOT);
    sc.blk_off = sc.blk_size + T;
    sc.frame_cbytes = sc.blk_off + T;
    if (!u.ztab) {
        std::vector<uint8_t> tab(zstd_tables_bytes());
        zstd_tables_host(tab.data());
        HIP_TRY(hipMalloc(&u.ztab, tab.size()));
        HIP_TRY(hipMemcpy(u.ztab, tab.data(), tab.size(), hipMemcpyHostToDevice));
    }
    launch_zstd_encode_blocks(sc, 1, u.ztab, u.stream);
    launch_scans(sc, 1, false, true, u.stream);
    HIP_TRY(hipMemcpyAsync(u.h_scalar, sc.frame_cbytes, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    const uint64_t total = 6ull + *reinterpret_cast<uint32_t *>(u.h_scalar);
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small (see rc_compress_bound)");
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_zstd_gather(sc, d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    *out_n = total;
    return RC_OK;
}

static int blosc_compress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    if (n >= (1ull << 31) - 16) return fail(RC_ERR_BAD_ARG, "rc_compress: a blosc1 chunk holds < 2 GiB");
    if (n == 0) {  // header only, "memcpyed"
        const uint8_t f[16] = {2, 1, 0x36, 8, 0, 0, 0, 0, 0, 0, 0, 0, 16, 0, 0, 0};
        if (dst_cap < 16) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small");
        HIP_TRY(hipMemcpy(dst, f, 16, is_device_ptr(dst) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        *out_n = 16;
        return RC_OK;
    }
    Scratch sc;
    sc.ntiles = (uint32_t)((n + TILE_BM - 1) / TILE_BM);
    sc.nb = n;
    sc.nb_stride = (uint64_t)sc.ntiles * TILE_BM;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src, sc.nb_stride - n + 16);
    if (r != RC_OK) return r;
    sc.bitmap = const_cast<uint8_t *>(d_src);
    const uint64_t T = sc.ntiles;
    r = ensure(u.w, u.w_cap, T * BLK_SLOT + T * 8 + 64);
    if (r != RC_OK) return r;
    sc.blk_slots = u.w;
    sc.blk_size = reinterpret_cast<uint32_t *>(u.w + T * BLK_SLOT);
    sc.blk_off = sc.blk_size + T;
    sc.frame_cbytes = sc.blk_off + T;
    launch_blosc_encode_blocks(sc, 1, u.stream);
    launch_scans(sc, 1, false, true, u.stream);
    HIP_TRY(hipMemcpyAsync(u.h_scalar, sc.frame_cbytes, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    const uint64_t total = 16ull + 4ull * T + *reinterpret_cast<uint32_t *>(u.h_scalar);
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small (see rc_compress_bound)");
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_blosc_gather(sc, d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    *out_n = total;
    return RC_OK;
}

RC_EXPORT int rc_compress(uint32_t scheme, uint32_t level, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap,
                          uint64_t *out_n)
{
    // level: LZ4 0 = the run parser, >= 1 = the event parser; zstd / blosc through this stateless seam: one effort (the ctx's zstd
    // encoder has the modelled form for level >= 1)
    if (!dst || !out_n || (!src && n)) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (scheme == RC_SCHEME_BLOSC_LZ4) {
        UtilScope util_scope;
        int r = util_scope.enter();
        if (r != RC_OK) return r;
        return blosc_compress(src, n, dst, dst_cap, out_n);
    }
    if (scheme != RC_SCHEME_LZ4 && scheme != RC_SCHEME_ZSTD)
        return fail(RC_ERR_UNSUPPORTED, "rc_compress: compression scheme not implemented on device");
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    return scheme == RC_SCHEME_LZ4 ? lz4_compress(src, n, dst, dst_cap, out_n, level) : zstd_compress(src, n, dst, dst_cap, out_n);
}
// blosc1 chunk with the LZ4 codec (what rc_compress(8) and python-blosc's cname='lz4' write): header and block table are
// walked on the host, the LZ4 blocks are decoded on the GPU into an image of the shuffled chunk, a second kernel unshuffles.
static int blosc_decompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    std::vector<uint8_t> hsrc;
    const uint8_t *h = src;
    if (is_device_ptr(src)) {
        hsrc.resize(n);
        HIP_TRY(hipMemcpy(hsrc.data(), src, n, hipMemcpyDeviceToHost));
        h = hsrc.data();
    }
    auto rd32 = [&](uint64_t p) { return (uint32_t)h[p] | ((uint32_t)h[p + 1] << 8) | ((uint32_t)h[p + 2] << 16) | ((uint32_t)h[p + 3] << 24); };
    if (n < 16 || h[0] != 2) return fail(RC_ERR_CORRUPT, "not a blosc1 chunk");
    const uint32_t flags = h[2], typesize = h[3] ? h[3] : 1;
    const uint64_t nbytes = rd32(4), blocksize = rd32(8), cbytes = rd32(12);
    if (cbytes != n || nbytes >= (1ull << 31)) return fail(RC_ERR_CORRUPT, "blosc1 header disagrees with the chunk length");
    *out_n = nbytes;
    if (nbytes > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small");
    if (nbytes == 0) return RC_OK;
    if (flags & 0x02) {  // memcpyed
        if (n != 16 + nbytes) return fail(RC_ERR_CORRUPT, "bad memcpyed blosc1 chunk");
        HIP_TRY(hipMemcpy(dst, src + 16, nbytes, hipMemcpyDefault));
        return RC_OK;
    }
    if ((flags >> 5) != 1) return fail(RC_ERR_UNSUPPORTED, "blosc1 chunk: only the LZ4 codec is decoded on device");
    if (blocksize == 0 || blocksize > nbytes) return fail(RC_ERR_CORRUPT, "bad blosc1 blocksize");
    const uint64_t nblocks = (nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * nblocks > n) return fail(RC_ERR_CORRUPT, "truncated blosc1 chunk");
    std::vector<Lz4Block> blks;
    std::vector<uint64_t> offs;
    std::vector<uint32_t> want;
    for (uint64_t b = 0; b < nblocks; ++b) {
        const uint64_t bsize = std::min<uint64_t>(blocksize, nbytes - b * blocksize);
        const bool leftover = bsize != blocksize;
        const bool split = !(flags & 0x10) && typesize <= 16 && blocksize / typesize >= 128 && !leftover;  // blosc.c blosc_d
        const uint32_t nsplits = split ? typesize : 1;
        const uint64_t neblock = bsize / nsplits;
        uint64_t pos = rd32(16 + 4 * b);
        for (uint32_t j = 0; j < nsplits; ++j) {
            if (pos + 4 > n) return fail(RC_ERR_CORRUPT, "blosc1 block table points outside the chunk");
            const uint32_t cs = rd32(pos);
            pos += 4;
            if (pos + cs > n || cs > neblock + neblock / 255 + 16) return fail(RC_ERR_CORRUPT, "blosc1 block exceeds the chunk");
            blks.push_back(Lz4Block{pos, cs, cs == neblock ? 1u : 0u});
            offs.push_back(b * blocksize + j * neblock);
            want.push_back((uint32_t)neblock);
            pos += cs;
        }
    }
    const uint32_t nb = (uint32_t)blks.size();
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src);
    if (r != RC_OK) return r;
    const uint64_t tab = (uint64_t)nb * sizeof(Lz4Block), szs = ((uint64_t)nb * 4 + 7) & ~7ull, ofs = (uint64_t)nb * 8;
    r = ensure(u.w, u.w_cap, tab + szs + ofs + 16);
    if (r != RC_OK) return r;
    r = ensure(u.b, u.b_cap, nbytes + 16);  // image of the shuffled chunk
    if (r != RC_OK) return r;
    Lz4Block *d_blks = reinterpret_cast<Lz4Block *>(u.w);
    uint32_t *d_sizes = reinterpret_cast<uint32_t *>(u.w + tab);
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(u.w + tab + szs);
    int *d_err = reinterpret_cast<int *>(u.w + tab + szs + ofs);
    HIP_TRY(hipMemcpyAsync(d_blks, blks.data(), tab, hipMemcpyHostToDevice, u.stream));
    HIP_TRY(hipMemcpyAsync(d_offs, offs.data(), ofs, hipMemcpyHostToDevice, u.stream));
    HIP_TRY(hipMemsetAsync(d_err, 0, 4, u.stream));
    launch_lz4_decode(d_src, d_blks, nb, d_sizes, nullptr, nullptr, ~0ull, 0, d_err, u.stream);  // sizes only: must equal the split size
    std::vector<uint32_t> sizes(nb);
    int err = 0;
    HIP_TRY(hipMemcpyAsync(sizes.data(), d_sizes, (uint64_t)nb * 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block inside the blosc1 chunk");
    for (uint32_t i = 0; i < nb; ++i)
        if (sizes[i] != want[i]) return fail(RC_ERR_CORRUPT, "blosc1 block decodes to the wrong size");
    uint32_t max_stored = 0;
    for (const Lz4Block &q : blks) if (q.raw) max_stored = std::max(max_stored, q.size);
    launch_lz4_decode(d_src, d_blks, nb, nullptr, d_offs, u.b, nbytes, 0, d_err, u.stream, max_stored);
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, nbytes);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_blosc_unshuffle(u.b, d_out, nbytes, (uint32_t)blocksize, typesize, (flags & 0x04) ? 4u : ((flags & 0x01) ? 1u : 0u), u.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, nbytes, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block inside the blosc1 chunk");
    return RC_OK;
}

// zstd frame of the subset the device decoders cover (rc_zstd_dec.h: everything rc_compress / the ctx write): the host walks
// the block headers and builds the tables, one lane decodes one block.  The decoded size is not in the frame: the last block
// is decoded "up to" a block's size and reports what it produced.
static int zstd_decompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    std::vector<uint8_t> hsrc;
    const uint8_t *h = src;
    if (is_device_ptr(src)) {
        hsrc.resize(n);
        HIP_TRY(hipMemcpy(hsrc.data(), src, n, hipMemcpyDeviceToHost));
        h = hsrc.data();
    }
    std::vector<ZdBlock> all, comp, raw;
    ZdTables T;
    uint64_t bound = 0;
    const int zr = zd_index_frame(h, 0, n, 0, TILE_BM, ~0ull, all, T, &bound);
    if (zr == ZD_FOREIGN) return fail(RC_ERR_UNSUPPORTED, "rc_decompress: zstd stream outside the device decoder's subset (use the stock decoder)");
    if (zr != ZD_OK) return fail(RC_ERR_CORRUPT, "malformed zstd frame");
    uint32_t raw_max = 0;
    for (const ZdBlock &b : all) {
        if (b.type == 2) { if (b.regen > 1024) return fail(RC_ERR_UNSUPPORTED, "rc_decompress: zstd block larger than the device decoder's rows"); comp.push_back(b); }
        else { raw.push_back(b); raw_max = std::max(raw_max, b.regen); }
    }
    // nothing is allocated or zeroed beyond what the caller's buffer justifies: a few KB of RLE blocks can announce gigabytes
    if (bound > dst_cap + 1024) { *out_n = bound; return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small (out_n = an upper bound of the decoded size)"); }
    uint32_t row = TILE_BM;
    for (const ZdBlock &b : comp) if (b.regen > (uint32_t)TILE_BM) row = 1024;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src);
    if (r != RC_OK) return r;
    const uint64_t sz_blk = (comp.size() + raw.size()) * sizeof(ZdBlock) + 64;
    if ((r = ensure(u.x[2], u.x_cap[2], sz_blk)) != RC_OK || (r = ensure(u.x[3], u.x_cap[3], sizeof(ZdTables) + 64)) != RC_OK ||
        (r = ensure(u.x[4], u.x_cap[4], 256)) != RC_OK || (r = ensure(u.x[1], u.x_cap[1], bound + 64)) != RC_OK)
        return r;
    if (!u.zd_predef) {
        std::vector<uint8_t> t(zd_tables_bytes());
        zd_predefined_tables(t.data());
        HIP_TRY(hipMalloc(&u.zd_predef, t.size()));
        HIP_TRY(hipMemcpy(u.zd_predef, t.data(), t.size(), hipMemcpyHostToDevice));
    }
    hipStream_t s = u.stream;
    ZdBlock *d_comp = reinterpret_cast<ZdBlock *>(u.x[2]), *d_raw = d_comp + comp.size();
    ZdFrameList *d_lists = reinterpret_cast<ZdFrameList *>(u.x[4]);   // [0] compressed, [1] stored / RLE
    uint64_t *d_base = reinterpret_cast<uint64_t *>(u.x[4] + 32);
    int *d_err = reinterpret_cast<int *>(u.x[4] + 48);
    uint32_t *d_prod = reinterpret_cast<uint32_t *>(u.x[4] + 56);
    const ZdFrameList lists[2] = {{d_comp, (uint32_t)comp.size(), 0}, {d_raw, (uint32_t)raw.size(), 0}};
    const uint64_t base0 = 0;
    const uint32_t none = 0xFFFFFFFFu;
    if (!comp.empty()) HIP_TRY(hipMemcpyAsync(d_comp, comp.data(), comp.size() * sizeof(ZdBlock), hipMemcpyHostToDevice, s));
    if (!raw.empty()) HIP_TRY(hipMemcpyAsync(d_raw, raw.data(), raw.size() * sizeof(ZdBlock), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(u.x[3], &T, sizeof T, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_lists, lists, sizeof lists, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_base, &base0, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(d_err, 0, 4, s));
    HIP_TRY(hipMemsetAsync(u.x[1], 0, bound + 64, s));   // the decoders store only what is not zero
    HIP_TRY(hipMemcpyAsync(d_prod, &none, 4, hipMemcpyHostToDevice, s));
    if (!comp.empty()) launch_block_decode(1, (int)row, d_src, d_lists, 1, (uint32_t)comp.size(), u.x[3], u.zd_predef, u.x[1], d_base, d_err, s, d_prod);
    launch_block_copy(d_src, d_lists + 1, 1, (uint32_t)raw.size(), raw_max, u.x[1], d_base, s);
    HIP_TRY(hipGetLastError());
    int err = 0;
    uint32_t prod = none;
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&prod, d_prod, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (err) return fail(RC_ERR_CORRUPT, "malformed zstd block");
    uint64_t total = bound;
    if (prod != none) {   // the flexible last block produced `prod` of the `regen` bytes it was given
        for (const ZdBlock &b : comp) if (b.flex) total = bound - (b.regen - prod);
    }
    *out_n = total;
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small");
    if (total) HIP_TRY(hipMemcpy(dst, u.x[1], total, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return RC_OK;
}

RC_EXPORT int rc_decompress(uint32_t scheme, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    if (!src || !out_n || (!dst && dst_cap)) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (scheme != RC_SCHEME_LZ4 && scheme != RC_SCHEME_BLOSC_LZ4 && scheme != RC_SCHEME_ZSTD)
        return fail(RC_ERR_UNSUPPORTED, "rc_decompress: compression scheme not implemented on device");
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    if (scheme == RC_SCHEME_ZSTD) return zstd_decompress(src, n, dst, dst_cap, out_n);   // RC_ERR_UNSUPPORTED for foreign frames: the
    return scheme == RC_SCHEME_LZ4 ? lz4_decompress(src, n, dst, dst_cap, out_n) : blosc_decompress(src, n, dst, dst_cap, out_n);   // caller's stock decoder
}
RC_EXPORT uint64_t rc_compress_bound(uint32_t scheme, uint64_t n)
{
    const uint64_t blocks = (n + rc::TILE_BM - 1) / rc::TILE_BM;
    if (scheme == RC_SCHEME_LZ4) return 7 + n + 4 * blocks + 4;
    if (scheme == RC_SCHEME_ZSTD) return 9 + n + 3 * blocks;
    if (scheme == RC_SCHEME_BLOSC_LZ4) return 16 + n + 8 * blocks;
    return 0;
}

// ---- synthetic stacks -------------------------------------------------------------------------------------------
RC_EXPORT int rc_synth_dark(int device_id, uint32_t seed, uint64_t n_pixels, uint16_t *dark_dev)
{
    if (!dark_dev) return fail(RC_ERR_BAD_ARG, "NULL argument");
    RC_ON_DEVICE(device_id);
    rc::launch_synth_dark(seed, n_pixels, dark_dev, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return RC_OK;
}
RC_EXPORT int rc_synth_frames_clustered(int device_id, uint32_t seed, uint32_t first_frame, uint32_t n_frames, uint32_t nx, uint32_t ny,
                                        uint32_t seed_ppm, const uint16_t *dark_dev, uint16_t *frames_dev)
{
    if (!dark_dev || !frames_dev || n_frames == 0 || nx == 0 || ny == 0) return fail(RC_ERR_BAD_ARG, "NULL / zero argument");
    if ((uint64_t)nx * ny > 0xFFFFFFFFull) return fail(RC_ERR_BAD_ARG, "rc_synth_frames_clustered: nx * ny must fit 32 bits");
    RC_ON_DEVICE(device_id);
    rc::launch_synth_frames_clustered(seed, first_frame, n_frames, nx, ny, seed_ppm, dark_dev, frames_dev, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return RC_OK;
}
RC_EXPORT int rc_synth_frames(int device_id, uint32_t seed, uint32_t first_frame, uint32_t n_frames, uint64_t n_pixels,
                              uint32_t sparsity_ppm, const uint16_t *dark_dev, uint16_t *frames_dev)
{
    if (!dark_dev || !frames_dev || n_frames == 0) return fail(RC_ERR_BAD_ARG, "NULL / zero argument");
    RC_ON_DEVICE(device_id);
    rc::launch_synth_frames(seed, first_frame, n_frames, n_pixels, sparsity_ppm, dark_dev, frames_dev, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return RC_OK;
}
