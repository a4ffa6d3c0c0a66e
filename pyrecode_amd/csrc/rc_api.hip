// rc_api.hip - the C ABI of librecode_hip.so (include/recode_hip.h): contexts, staging, entry points.
// No CPU implementation lives here: every compute entry point runs HIP kernels or returns RC_ERR_DEVICE.
#include "rc_host.h"

thread_local std::string g_last_error;
WorkerPool *g_pool = new WorkerPool;
Util g_utils[RC_MAX_DEV];
thread_local Util *t_util = nullptr;


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
    hipStream_t pstream_b = nullptr;      // two chains (pipelined, level 2): the second stage of the batches on scratch set 1
    hipStream_t last_ps = nullptr;        // the stream the most recent batch's second stage went to
    bool two_chains = false;
    hipEvent_t ev_red[2] = {}, ev_post[2] = {}, ev_in[2] = {};
    bool post_pending[2] = {false, false};
    bool pipelined = false;
    bool thr_set = false;
    bool keep_bitmap = true;  // also store the raw binary maps when a device codec is active (rc_get_binary_map)
    uint32_t src_bytes = 2;   // bytes per source pixel: 2 (uint16 frames and dark), 1 (uint8) or 4 (uint32): rc_ctx_set_source_bytes
    uint32_t *thr32 = nullptr;   // uint32 sources: the threshold frame (sc.thr is the uint16 one)
    uint32_t last_n = 0;
    // staging for host callers
    uint8_t *d_frames = nullptr;  uint64_t d_frames_cap = 0;
    uint8_t *d_out = nullptr;     uint64_t d_out_cap = 0;
    uint8_t *d_dark = nullptr;    uint64_t d_dark_cap = 0;
    uint64_t *d_rec_off = nullptr;
    uint32_t *d_md = nullptr;
    void *d_ztab = nullptr;               // zstd FSE tables (emit == 1)
    // modelled zstd (compression_level >= 1): tables fitted to a sample of the ctx's first batch (rc_zstd_model.h)
    bool modelled = false, model_ready = false;
    rc::ZstdModel *d_model = nullptr, *h_model = nullptr;
    rc::ZstdSample *d_sample = nullptr, *h_sample = nullptr;
    uint32_t l2_sum = 0;                  // L2_statistics: 0/1 max, 2 sum
    rc::u32x2 *d_l2_node[2] = {};         // level 2: the labelling stage's nodes (rc_l2.hip) - one workspace per chain of second stages
    uint16_t *d_l2_base[2] = {};          // ... and its directory of rank bases
    rc::BatchStatus *h_status = nullptr;  // pinned: [0] most recent batch, [1] first failed batch since the last sync
    rc::BatchStatus *d_first_err = nullptr;
    uint32_t batch_seq = 0;               // batches enqueued since the last rc_ctx_sync
    // the previous batch's output buffers (two chains: a caller that hands consecutive batches the SAME buffers is serialised, see enqueue_batch)
    const uint8_t *prev_out = nullptr, *prev_rec = nullptr, *prev_md = nullptr;
    uint64_t prev_out_cap = 0;
    uint32_t prev_n = 0;
    hipEvent_t ev[5] = {};
    float stage_ms[5] = {};
    // optional per-enqueue stage events for the asynchronous path (rc_ctx_set_profiling)
    bool profiling = false;
    uint32_t prof_every = 1, prof_phase = 0;   // events around every prof_every-th batch (rc_ctx_set_profiling(ctx, k))
    bool profile_all = getenv("RC_PROFILE_ALL_STAGES") != nullptr;
    // host streaming form (rc_pipe_*): per slot device buffers, pinned metadata, events
    struct PipeSlot {
        uint8_t *d_in = nullptr;
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
    case RC_ERR_WORKSPACE: return "out of device memory for the ctx's workspace";
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
    return (scheme == RC_SCHEME_LZ4 || scheme == RC_SCHEME_ZSTD || scheme == RC_SCHEME_BLOSC_LZ4 || scheme == RC_SCHEME_ZLIB_DEVICE) ? 1 : 0;
}

// ---- seam 1 --------------------------------------------------------------------------------------------------
static int alloc_set(rc_ctx *c, rc::Scratch &sc)
{
    using namespace rc;
    const uint64_t B = c->max_batch, T = sc.ntiles;
    HIP_TRY(hipMalloc((void **)&sc.bitmap, B * sc.nb_stride + 64));  // + slack: k_gather reads whole 16-byte pieces
    HIP_TRY(hipMalloc((void **)&sc.tile_cnt, B * T * 4));
    HIP_TRY(hipMalloc((void **)&sc.tile_off, B * T * 4));
    HIP_TRY(hipMalloc((void **)&sc.tile_next, B * T * 4));
    HIP_TRY(hipMalloc((void **)&sc.frame_nnz, B * 4));
    HIP_TRY(hipMalloc((void **)&sc.frame_cbytes, B * 4));
    HIP_TRY(hipMalloc((void **)&sc.scan_part, B * ((T + 255) / 256) * 32));   // (a row of partials per scan segment; room for segments down to 256 tiles: RC_SCAN_T experiments)
    HIP_TRY(hipMalloc((void **)&sc.status, sizeof(BatchStatus)));
    if (c->level != 3) HIP_TRY(hipMalloc((void **)&sc.pix_slots, B * T * TILE_PX * 2 + 64));
    if (c->emit != 0) {
        if (c->level == 1 && !RC_KNOB("RC_NO_COMBINED_SLOTS")) {   // combined slots (rc_launch.h, Scratch::comb); the knob: A/B runs
            sc.comb = c->emit == RC_SCHEME_ZSTD ? 2u : 1u;
            if (const char *e = RC_KNOB("RC_COMB_MODE")) {   // (A/B runs; zstd's blocks still grow behind the reduce kernel: form 1 would put residuals in their way)
                const uint32_t m = (uint32_t)atoi(e);
                if (m <= 2 && !(c->emit == RC_SCHEME_ZSTD && m == 1)) sc.comb = m;
            }
            sc.blk_stride = 1536;                                  // 12 lines: the block image (<= 5) + 7 or more lines of residuals
        }
        HIP_TRY(hipMalloc((void **)&sc.blk_slots, B * T * (uint64_t)sc.blk_stride + 256));
        HIP_TRY(hipMalloc((void **)&sc.blk_size, B * T * 4));
        HIP_TRY(hipMalloc((void **)&sc.blk_off, B * T * 4));
        if (c->emit == rc::EMIT_DEFLATE) {   // the zlib streams' Adler-32: per-tile partials of the map, per-frame sums (rc_deflate_block.h, k_gather)
            HIP_TRY(hipMalloc((void **)&sc.blk_aux, B * T * 4));
            HIP_TRY(hipMalloc((void **)&sc.zl_acc, B * 32));
            HIP_TRY(hipMemset(sc.zl_acc, 0, B * 32));
        }
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

static int ctx_alloc(rc_ctx *c)
{
    using namespace rc;
    const uint64_t B = c->max_batch;
    RC_ON_DEVICE(c->device);
    {
        // Experiment knob: RC_PSTREAM_CUS=n with RC_RSTREAM_EXCL=1 - the ctx's own stream (the reduce kernel, when the caller sets no stream) is
        // confined to the CUs the second stage is NOT confined to: the two stages share no CU at all.
        const char *e = RC_KNOB("RC_PSTREAM_CUS");
        const int ncu = e ? atoi(e) : 0;
        if (ncu > 0 && ncu < 256 && RC_KNOB("RC_RSTREAM_EXCL")) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = ncu; i < 256; ++i) mask[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&c->own_stream, 8, mask) != hipSuccess) { (void)hipGetLastError(); c->own_stream = nullptr; }
        }
    }
    if (!c->own_stream) HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    // (a high-priority second-stage stream was measured: no gain with LZ4 or zstd, 2 % slower at 11520x8184 - tools/ab_bench.sh)
    if (const char *pe = RC_KNOB("RC_PSTREAM_PRIO")) {   // experiment knob: the second stage on a high-priority stream (2: a low-priority one)
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&c->pstream_all, hipStreamNonBlocking, atoi(pe) == 2 ? lo : hi));
    } else
        HIP_TRY(hipStreamCreateWithFlags(&c->pstream_all, hipStreamNonBlocking));
    {
        // Experiment knob, off by default.  In pipelined mode the second stage runs next to the following batch's reduce
        // kernel; its waves (80 VGPRs, latency bound) settle on every SIMD and push out one of the three reduce waves
        // there (168 VGPRs each).  RC_PSTREAM_CUS=n confines the second stage to the first n CU-mask bits.  Measured on
        // bench.py, same box: LZ4 123.7 k frames/s unmasked, 126.2 k with n = 104 (96: 125.6-127.6 k, 128: 128.5 k on a
        // faster box, 64: no gain, spread-out masks: worse) - but zstd, whose second stage also carries the FSE kernel,
        // drops from 112 k to 103 k: the confined stage becomes the longer one.  +2 % on one codec does not pay for that.
        const char *e = RC_KNOB("RC_PSTREAM_CUS");
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
    {
        // (the second chain's stream: created at another priority so that it gets a hardware queue of its own - a plain third stream shared
        // the reduce stream's queue on the default four and cost every configuration 10 %, profiles/r05_exp23_two_chains_shared_queue.log)
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&c->pstream_b, hipStreamNonBlocking, hi));
    }
    c->pstream = c->pstream_all;
    c->last_ps = c->pstream;
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
    // Two chains (level 2): the second stage of a level-2 batch - directory, links, statistics, emit, then the level-1 stages - is a chain of
    // latency-bound kernels about as long as the reduce kernel itself; on ONE stream it paces the batches (step 0.70 ms against a reduce kernel
    // of 0.58).  The batches on scratch set 1 take a second stream and a second workspace (nodes + directory), so two consecutive
    // batches' chains overlap each other as well as the reduce kernels: +4 % at 1 % of the pixels set, +9 % on clustered events, nothing for
    // level 1 whose chain is a quarter of its reduce kernel (profiles/r05_exp24_two_chains.log).  Batches still COMPLETE in order.
    c->two_chains = false;   // (the second chain's workspace comes with the first rc_ctx_set_pipelined(ctx, 1): l2_second_chain)
    if (c->level == 2) {
        // level 2 (rc_l2.hip): a node {parent, accumulator} for every pixel of a batch (8 bytes each: 8.6 GB for 64 frames of 4096^2, of
        // which only the entries of set pixels with neighbours are ever touched), at rest - zero - between batches, and the directory of
        // rank bases (2 bytes per 64 pixels) - sized by the geometry, so no batch can exceed it.  One workspace per chain: a ctx that is
        // never pipelined has one chain and one workspace.
        const uint64_t ids = (uint64_t)c->sc.ntiles * rc::TILE_PX;
        HIP_TRY(hipMalloc((void **)&c->d_l2_node[0], B * ids * 8));
        HIP_TRY(hipMemset(c->d_l2_node[0], 0, B * ids * 8));
        HIP_TRY(hipMalloc((void **)&c->d_l2_base[0], B * (uint64_t)c->sc.ntiles * 64 * 2));   // (written by k_l2_dir before anything reads it)
        for (int i = 0; i < 2; ++i) {
            Scratch &set = c->sets[i];
            set.l2_node = c->d_l2_node[0]; set.l2_ids_per_frame = ids; set.l2_base = c->d_l2_base[0];
        }
    }
    c->sc = c->sets[0];
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
    if (nx == 0 || ny == 0 || max_batch == 0 || max_batch > 65535u || op_mode > 1) {   // (a batch's frames are a launch's grid.y)
        *status = fail(RC_ERR_BAD_ARG, "nx, ny, max_batch must be > 0, max_batch <= 65535 and op_mode 0 or 1");
        return nullptr;
    }
    if ((uint64_t)nx * ny >= (1ull << 32)) {
        *status = fail(RC_ERR_BAD_ARG, "nx*ny must be < 2^32");
        return nullptr;
    }
    // A record may be as large as its raw frame (recode_writer.py:565-566), the format keeps every size of a record in a u32 metadata field
    // (structures.py:18-46) and k_gather places bytes at u32 offsets inside a record: a raw frame of 4 GiB or more cannot be represented
    // (uint16 sources: nx * ny < 2^31, e.g. 46 340^2; the reference's own C type holds nx, ny in 16 bits each, pyrecode.cpp:21-22, which
    // would allow 65 535^2).  rc_ctx_set_source_bytes checks again for the pixel size it is given.
    if ((uint64_t)nx * ny * 2 >= (1ull << 32)) {
        *status = fail(RC_ERR_UNSUPPORTED, "a raw frame of 4 GiB or more (nx*ny*bytes_per_pixel >= 2^32) does not fit the format's u32 size fields");
        return nullptr;
    }
    if (reduction_level < 1 || reduction_level > 3) {
        *status = fail(RC_ERR_UNSUPPORTED, "reduction_level 4 (centroiding) is not implemented on device");
        return nullptr;
    }
    if (reduction_level == 2 && nx > 65535u) {   // (the reference's C type holds nx in 16 bits, pyrecode.cpp:21-22; the labelling stage is tested up to there)
        *status = fail(RC_ERR_UNSUPPORTED, "reduction_level 2 needs nx <= 65535");
        return nullptr;
    }
    if (src_bit_depth < 1 || src_bit_depth > 32) {   // (<= 8: the reference's source dtype is uint8, > 16: uint32 - rc_ctx_set_source_bytes)
        *status = fail(RC_ERR_UNSUPPORTED, "source_bit_depth must be 1..32 (uint8 / uint16 / uint32 source frames)");
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
    sc.guarded_loads = getenv("RC_REDUCE_GUARDED_LOADS") != nullptr;   // (tests; read once per ctx)
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
    if (c->pstream_b) (void)hipStreamSynchronize(c->pstream_b);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (rc::Scratch &sc : c->sets) {
        void *per_set[] = {sc.bitmap, sc.pix_slots, sc.tile_cnt, sc.tile_off, sc.tile_next, sc.blk_slots, sc.blk_size, sc.blk_aux, sc.zl_acc,
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
    void *bufs[] = {c->sc.thr, c->thr32, c->d_first_err, c->d_frames, c->d_out, c->d_dark, c->d_rec_off,
                    c->d_md, c->d_ztab, c->d_model, c->d_sample, c->d_l2_node[0], c->d_l2_node[1], c->d_l2_base[0], c->d_l2_base[1]};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    hipEvent_t sync_ev[] = {c->ev_red[0], c->ev_red[1], c->ev_post[0], c->ev_post[1], c->ev_in[0], c->ev_in[1]};
    for (hipEvent_t e : sync_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->pstream_all) (void)hipStreamDestroy(c->pstream_all);
    if (c->pstream_masked) (void)hipStreamDestroy(c->pstream_masked);
    if (c->pstream_b) (void)hipStreamDestroy(c->pstream_b);
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

RC_EXPORT int rc_set_threshold(rc_ctx *c, const void *thr)
{
    if (!c || !thr) return fail(RC_ERR_BAD_ARG, "ctx / thr is NULL");
    if (c->depth > 16 && c->src_bytes != 4) return fail(RC_ERR_BAD_ARG, "src_bit_depth > 16 means uint32 sources: rc_ctx_set_source_bytes(ctx, 4) first");
    RC_ON_DEVICE(c->device);
    // (uint16 thresholds for uint16 AND uint8 sources - the device keeps them as uint16 -, uint32 ones for uint32 sources)
    HIP_TRY(hipMemcpyAsync(c->src_bytes == 4 ? (void *)c->thr32 : (void *)c->sc.thr, thr, c->sc.N * (c->src_bytes == 4 ? 4 : 2),
                           is_device_ptr(thr) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->thr_set = true;
    return RC_OK;
}

RC_EXPORT int rc_set_dark(rc_ctx *c, const void *dark, int64_t epsilon)
{
    if (!c || !dark) return fail(RC_ERR_BAD_ARG, "ctx / dark is NULL");
    RC_ON_DEVICE(c->device);
    const void *src = dark;
    const uint64_t bytes = c->sc.N * c->src_bytes;
    if (!is_device_ptr(dark)) {
        int r = ensure(c->d_dark, c->d_dark_cap, bytes);
        if (r != RC_OK) return r;
        HIP_TRY(hipMemcpyAsync(c->d_dark, dark, bytes, hipMemcpyHostToDevice, c->stream));
        src = c->d_dark;
    }
    if (c->depth > 16 && c->src_bytes != 4) return fail(RC_ERR_BAD_ARG, "src_bit_depth > 16 means uint32 sources: rc_ctx_set_source_bytes(ctx, 4) first");
    if (c->src_bytes == 4) rc::launch_threshold32(static_cast<const uint32_t *>(src), epsilon, c->sc.N, c->thr32, c->stream);
    else rc::launch_threshold(src, epsilon, c->sc.N, c->sc.thr, c->stream, c->src_bytes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->thr_set = true;
    return RC_OK;
}

// Source pixels of one byte (source_bit_depth <= 8: the reference's map_dtype gives uint8 frames and dark, misc.py:41-49).  Before
// rc_set_dark and the first batch; everything behind the frame loads is the same path (values, packing, codecs, records).
RC_EXPORT int rc_ctx_set_source_bytes(rc_ctx *c, uint32_t bytes_per_pixel)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    if (bytes_per_pixel != 1 && bytes_per_pixel != 2 && bytes_per_pixel != 4) return fail(RC_ERR_BAD_ARG, "rc_ctx_set_source_bytes: 1 (uint8), 2 (uint16) or 4 (uint32)");
    if (bytes_per_pixel == 1 && c->depth > 8) return fail(RC_ERR_BAD_ARG, "rc_ctx_set_source_bytes: uint8 sources need src_bit_depth <= 8");
    if (bytes_per_pixel == 2 && c->depth > 16) return fail(RC_ERR_BAD_ARG, "rc_ctx_set_source_bytes: uint16 sources need src_bit_depth <= 16");
    if (c->batch_seq || c->thr_set) return fail(RC_ERR_BAD_ARG, "rc_ctx_set_source_bytes: call before rc_set_dark / rc_set_threshold and the first batch");
    if (c->sc.N * bytes_per_pixel >= (1ull << 32))
        return fail(RC_ERR_UNSUPPORTED, "rc_ctx_set_source_bytes: a raw frame of 4 GiB or more (nx*ny*bytes_per_pixel >= 2^32) does not fit the format's u32 size fields");
    if (bytes_per_pixel == 4) {
        // uint32 sources (source_bit_depth > 16, misc.py:41-49): rc_reduce32.hip.  Levels 1 and 3; the residual fields are depth bits wide, or
        // the values' four raw bytes when the depth is a multiple of 8 (`.tobytes()` of a uint32 array, recode_writer.py:463-464: 32 and 24
        // alike); zstd takes the fast encoder (the modelled one is fitted inside the uint16 kernel).
        if (c->depth <= 16) return fail(RC_ERR_BAD_ARG, "rc_ctx_set_source_bytes: uint32 sources are what source_bit_depth > 16 means (misc.py:41-49)");
        if (c->level == 2) return fail(RC_ERR_UNSUPPORTED, "rc_ctx_set_source_bytes: reduction level 2 is not implemented for uint32 sources");
        if (c->emit == rc::EMIT_DEFLATE) return fail(RC_ERR_UNSUPPORTED, "rc_ctx_set_source_bytes: the device DEFLATE encoder is not implemented for uint32 sources (use RC_SCHEME_ZLIB: the host's zlib)");
        RC_ON_DEVICE(c->device);
        // Everything that can fail happens first, into temporaries; the ctx changes only once all of it has succeeded (a failed
        // allocation leaves the uint16 ctx as it was).
        uint32_t *thr32 = nullptr;
        uint16_t *slots[2] = {nullptr, nullptr};
        bool ok = hipMalloc((void **)&thr32, c->sc.N * 4) == hipSuccess;
        for (int k = 0; ok && k < 2 && c->level != 3; ++k)
            ok = hipMalloc((void **)&slots[k], (uint64_t)c->max_batch * c->sets[k].ntiles * rc::TILE_PX * 4 + 64) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (thr32) (void)hipFree(thr32);
            for (uint16_t *p : slots) if (p) (void)hipFree(p);
            return fail(RC_ERR_DEVICE, "rc_ctx_set_source_bytes: out of device memory for the uint32 threshold / residual slots");
        }
        if (c->depth % 8 == 0) c->depth = 32;
        c->modelled = false;
        c->thr32 = thr32;
        for (int k = 0; k < 2; ++k) {
            rc::Scratch *set = &c->sets[k];
            set->pix_slot_bytes = rc::TILE_PX * 4;
            set->comb = 0;
            if (c->level != 3) {
                (void)hipFree(set->pix_slots);
                set->pix_slots = slots[k];
            }
            // the modelled zstd encoder's residual-stream scratch (alloc_set sized it for uint16 sources): not used by this path
            void **unused[] = {(void **)&set->pixraw, (void **)&set->pix_chunks, (void **)&set->chunk_size, (void **)&set->chunk_off, (void **)&set->frame_pbytes};
            for (void **q : unused) { if (*q) (void)hipFree(*q); *q = nullptr; }
            set->pixraw_stride = 0; set->nchunk_max = 0;
        }
        c->sc = c->sets[0];
    }
    c->src_bytes = bytes_per_pixel;
    return RC_OK;
}
RC_EXPORT uint32_t rc_ctx_source_bytes(const rc_ctx *c) { return c ? c->src_bytes : 0; }

RC_EXPORT uint64_t rc_out_capacity(const rc_ctx *c, uint32_t n) { return c ? (uint64_t)n * c->sc.N * c->src_bytes : 0; }

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
static int fit_model(rc_ctx *c, const void *frames_dev, uint32_t n)
{
    using namespace rc;
    hipStream_t s = c->stream;
    const Scratch &sc = c->sets[c->cur];
    const uint32_t ns = n < 2 ? n : 2;
    for (int k = 0; k < 2; ++k)
        if (c->post_pending[k]) HIP_TRY(hipStreamWaitEvent(s, c->ev_post[k], 0));
    HIP_TRY(hipMemsetAsync(c->d_sample, 0, sizeof(ZstdSample), s));
    launch_reduce(sc, frames_dev, ns, c->level == 3 ? 3u : 1u, 1u, true, c->depth, s, nullptr, c->src_bytes);   // (with the raw maps: the sample counts all their bytes)
    launch_zstd_sample(sc, ns, c->level == 1, c->depth, c->d_sample, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_sample, c->d_sample, sizeof(ZstdSample), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // compression_level -> how much size the faster block form of the binary maps may cost (rc_zstd_model.h): the reference hands the
    // level to libzstd (recode_writer.py:175-178), where 1 is the fast end too.  Level 1: up to 5 % of the binary-map stream, level 2: 2 %,
    // from 3 on the smaller form always.
    zstd_model_from_sample(c->h_sample, c->h_model, c->clevel == 1 ? 50u : (c->clevel == 2 ? 20u : 0u));
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

static int enqueue_batch(rc_ctx *c, const void *frames_dev, uint32_t n, uint32_t first_frame_id, uint8_t *out_dev,
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
    rp.packed_slots = 1u;   // (level 2: k_l2_emit leaves its statistics as tile-local packed streams, like level-1 residuals)
    rp.frame_bytes = c->sc.N * c->src_bytes;   // a record may not exceed the raw frame (recode_writer.py:565-566)
    hipEvent_t *ev = nullptr;
    if (timed) ev = c->ev;
    else if (c->profiling && (c->prof_phase++ % c->prof_every) == 0) {
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
    // (two chains: consecutive batches' second stages on two streams - they overlap each other as well as the reduce kernels)
    const bool two = c->pipelined && c->two_chains && c->pstream == c->pstream_all;
    hipStream_t ps = two && k == 1 ? c->pstream_b : c->pstream;
    c->last_ps = ps;
    // the batch two calls ago must have left this set.  (Round 5 tried to skip the wait when hipEventQuery says the event has completed -
    // it always has, in steady state: no gain in the step, 9 us more host time per call.)
    if (c->post_pending[k]) HIP_TRY(hipStreamWaitEvent(s, c->ev_post[k], 0));
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
    const uint32_t codec = c->modelled ? 3u : (c->emit == RC_SCHEME_LZ4 && c->clevel != 0 ? 4u : (c->emit == EMIT_DEFLATE ? 5u : c->emit));
    if (c->src_bytes == 4) {
        // uint32 sources (rc_reduce32.hip): reduce + pack with the codec's block encoder inside the kernel, as in the uint16 path (zstd in its
        // fast form)
        launch_reduce32(sc, static_cast<const uint32_t *>(frames_dev), c->thr32, n, c->level, c->depth, s,
                        c->emit == RC_SCHEME_LZ4 ? (c->clevel != 0 ? 4u : 2u) : (c->emit == RC_SCHEME_BLOSC_LZ4 ? 8u : (c->emit == RC_SCHEME_ZSTD ? 1u : 0u)),
                        c->keep_bitmap || c->emit == 0);
    } else
        launch_reduce(sc, frames_dev, n, c->level, codec, c->keep_bitmap || c->emit == 0, c->depth, s, tail, c->src_bytes);
    // every event costs a few microseconds of stream time: the asynchronous path records only the ones it needs
    // (start, end of the reduce kernel, end of the batch) unless RC_PROFILE_ALL_STAGES is set
    const bool all_ev = ev && (timed || c->profile_all);
    // (one event behind the reduce kernel: every packet on this stream is a few microseconds between two reduce kernels)
    hipEvent_t red = ev ? ev[1] : c->ev_red[k];
    HIP_TRY(hipEventRecord(red, s));
    HIP_TRY(hipStreamWaitEvent(ps, red, 0));
    if (c->level == 2) launch_l2(sc, n, c->nx, c->l2_sum, c->depth, ps);   // the tiles' raw values -> their components' statistics (rc_l2.hip)
#ifdef RC_DEV_SKIP   // development builds only (tools/build_def.sh): leave second-stage kernels out (WRONG records) to see what each costs the
                     // reduce kernel running next to it - bits: 1 FSE, 2 scans, 4 residual Huffman chain, 8 layout, 16 assemble, 32 gather
    static const unsigned skip = getenv("RC_DEV_SKIP_BITS") ? (unsigned)atoi(getenv("RC_DEV_SKIP_BITS")) : 0u;
#else
    constexpr unsigned skip = 0;
#endif
    const bool lits_only = c->modelled && (c->h_model->valid & ZM_LITS_ONLY);   // dense maps: no block has sequences, nothing for the FSE chain to do
    if (c->emit == RC_SCHEME_ZSTD && !(skip & 1) && !lits_only) launch_zstd_fse(sc, n, fitted_seq ? (const void *)&c->d_model->seq : c->d_ztab, fitted_seq, ps);
    if (all_ev) HIP_TRY(hipEventRecord(ev[2], ps));
    if (!(skip & 2)) launch_scans(sc, n, c->level != 3, c->emit != 0, ps);
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
    if (two && c->post_pending[k ^ 1]) {
        // Two chains run the second stages of batches i and i + 1 at the same time, which is why include/recode_hip.h asks a pipelined caller for
        // two sets of output buffers.  A caller that hands this batch a buffer the previous batch is still writing (rounds 1-4 tolerated that: one
        // second-stage stream serialised them) gets that order back instead of torn records: this batch's layout waits for the other chain.
        auto overlap = [](const uint8_t *a, uint64_t an, const uint8_t *b, uint64_t bn) { return a && b && a < b + bn && b < a + an; };
        const uint8_t *rec8 = reinterpret_cast<const uint8_t *>(rec_off_dev), *md8 = reinterpret_cast<const uint8_t *>(md_dev);
        if (overlap(out_dev, out_cap, c->prev_out, c->prev_out_cap) || overlap(rec8, (uint64_t)(n + 1) * 8, c->prev_rec, (uint64_t)(c->prev_n + 1) * 8) ||
            overlap(md8, (uint64_t)n * 12, c->prev_md, (uint64_t)c->prev_n * 12))
            HIP_TRY(hipStreamWaitEvent(ps, c->ev_post[k ^ 1], 0));
    }
    c->prev_out = out_dev; c->prev_out_cap = out_cap; c->prev_n = n;
    c->prev_rec = reinterpret_cast<const uint8_t *>(rec_off_dev); c->prev_md = reinterpret_cast<const uint8_t *>(md_dev);
    if (!(skip & 8)) launch_layout(sc, rp, n, out_cap, rec_off_dev, md_dev, ps);
    if (!(skip & 16)) launch_assemble(sc, rp, n, out_dev, rec_off_dev, c->batch_seq, ps);
    ++c->batch_seq;
    if (pix_huff && !(skip & 32)) launch_pix_gather(sc, n, c->depth, 16, out_dev, rec_off_dev, ps);
    if (ev) HIP_TRY(hipEventRecord(ev[4], ps));
    if (two && c->post_pending[k ^ 1]) HIP_TRY(hipStreamWaitEvent(ps, c->ev_post[k ^ 1], 0));   // batches still COMPLETE in order
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

RC_EXPORT int rc_reduce_compress_batch_async(rc_ctx *c, const void *frames_dev, uint32_t n, uint32_t first_frame_id,
                                             uint8_t *out_dev, uint64_t out_cap, uint64_t *rec_offsets_dev, uint32_t *md_dev)
{
    int r = check_batch_args(c, frames_dev, n, out_dev, rec_offsets_dev, md_dev);
    if (r != RC_OK) return r;
    RC_ON_DEVICE(c->device);
    return enqueue_batch(c, frames_dev, n, first_frame_id, out_dev, out_cap, rec_offsets_dev, md_dev, false);
}

// Level 2, first rc_ctx_set_pipelined(ctx, 1): the second chain's workspace (another 8 bytes per pixel and frame of the batch).  A GPU that
// cannot hold it keeps ONE chain - a few per cent slower (rc_api.hip, "two chains"), not an error.  The streams are drained when this runs.
static void l2_second_chain(rc_ctx *c)
{
    if (c->level != 2 || c->two_chains || RC_KNOB("RC_ONE_CHAIN")) return;
    const uint64_t B = c->max_batch, ids = (uint64_t)c->sc.ntiles * rc::TILE_PX;
    if (!c->d_l2_node[1]) {
        bool ok = hipMalloc((void **)&c->d_l2_node[1], B * ids * 8) == hipSuccess && hipMemset(c->d_l2_node[1], 0, B * ids * 8) == hipSuccess &&
                  hipMalloc((void **)&c->d_l2_base[1], B * (uint64_t)c->sc.ntiles * 64 * 2) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (c->d_l2_node[1]) (void)hipFree(c->d_l2_node[1]);
            if (c->d_l2_base[1]) (void)hipFree(c->d_l2_base[1]);
            c->d_l2_node[1] = nullptr; c->d_l2_base[1] = nullptr;
            return;
        }
    }
    c->sets[1].l2_node = c->d_l2_node[1];
    c->sets[1].l2_base = c->d_l2_base[1];
    if (c->last == 1) c->sc = c->sets[1];
    c->two_chains = true;
}

RC_EXPORT int rc_ctx_set_pipelined(rc_ctx *c, int on)
{
    if (!c) return fail(RC_ERR_BAD_ARG, "ctx is NULL");
    RC_ON_DEVICE(c->device);
    HIP_TRY(hipStreamSynchronize(c->pstream));  // the second stage changes streams: drain the old one first
    HIP_TRY(hipStreamSynchronize(c->pstream_b));
    if (on) l2_second_chain(c);
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
    HIP_TRY(hipStreamSynchronize(c->pstream_b));
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
    if (c->h_status[1].total != 0) c->h_status[1] = rc::first_err_decode(c->h_status[1].total);   // (k_gather leaves a key: rc_device.h)
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

RC_EXPORT int rc_reduce_compress_batch(rc_ctx *c, const void *frames, uint32_t n, uint32_t first_frame_id, uint8_t *out,
                                       uint64_t out_cap, uint64_t *rec_offsets, uint32_t *md)
{
    int r = check_batch_args(c, frames, n, out, rec_offsets, md);
    if (r != RC_OK) return r;
    RC_ON_DEVICE(c->device);
    const uint64_t frame_bytes = c->sc.N * c->src_bytes;
    const void *fdev = frames;
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
    HIP_TRY(hipMalloc((void **)&p.d_in, B * c->sc.N * c->src_bytes + 64));
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

RC_EXPORT int rc_pipe_submit(rc_ctx *c, uint32_t slot, const void *frames_host, uint32_t n, uint32_t first_frame_id)
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
        HIP_TRY(hipStreamSynchronize(c->pstream_b));
        l2_second_chain(c);
        c->pipelined = true;
        c->pstream = c->pstream_masked ? c->pstream_masked : c->pstream_all;
    }
    // Page-locked (or registered) frames are read by the reduce kernel IN PLACE, over the link: every frame byte is needed
    // exactly once, by wide nontemporal loads, so a copy into device memory first would only add a pass (and the copy
    // engines moved 26-31 GB/s here where the kernel's own reads move what the link gives).  RC_PIPE_COPY=1 forces the copy.
    const void *fdev = nullptr;
    static const bool force_copy = RC_KNOB("RC_PIPE_COPY") != nullptr;
    if (!force_copy) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, frames_host) == hipSuccess) {
            if (a.type == hipMemoryTypeHost && a.devicePointer) fdev = a.devicePointer;
        } else (void)hipGetLastError();
    }
    p.zero_copy = fdev != nullptr;
    if (!fdev) {
        HIP_TRY(hipMemcpyAsync(p.d_in, frames_host, (uint64_t)n * c->sc.N * c->src_bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(p.ev_h2d, c->copy_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, p.ev_h2d, 0));
        fdev = p.d_in;
    }
    int r = enqueue_batch(c, fdev, n, first_frame_id, p.d_out, rc_out_capacity(c, c->max_batch), p.d_rec, p.d_md, false);
    if (r != RC_OK) return r;
    p.has_val = c->val_gap != 0;
    if (p.has_val) {   // validation frames of this batch: the dose-rate count, from the frames the reduce kernel has just read
        rc::launch_roi_components(fdev, c->src_bytes == 4 ? (const void *)c->thr32 : (const void *)c->sc.thr, c->sc.N, c->nx, n, first_frame_id, c->val_gap, c->val_x0, c->val_y0, c->val_w, c->val_h, p.d_val, c->stream, c->src_bytes);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(p.h_val, p.d_val, (uint64_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(p.ev_val, c->stream));
    }
    if (p.zero_copy) HIP_TRY(hipEventRecord(p.ev_h2d, c->stream));   // "input consumed" = the reduce kernel (and the count) have run
    hipStream_t ps = c->last_ps;   // carries the batch's assembly: the metadata follows it
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
    c->prof_every = on > 1 ? (uint32_t)on : 1u;
    c->prof_phase = 0;
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
