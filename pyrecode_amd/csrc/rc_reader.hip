// rc_reader.hip - the batched reader behind rc_expand_frames / rc_expand_frames_submit / _wait (include/recode_hip.h, seam 3):
// host walk of the stored frames' block headers on a small thread pool, block lists in page-locked memory, then the device
// decoders (rc_zstd_dec.hip) and the sparse expand (rc_expand.hip) - n frames, both streams, one call, no host round trip in between.
// Replaces ReCoDeReader._get_frame_sparse (pyrecode/recode_reader.py:379-471) for n frames at once.
#include "rc_host.h"

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
// coo: the output is not uint64 triplets but the three arrays of a COO matrix - int32 rows[cap] | int32 columns[cap] | uint16 values[cap]
// (k_expand_emit_b<true>): 10 bytes per set pixel and capacity instead of 24.
static int expand_run(uint32_t slot, bool submit_only, uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t level, uint32_t op_mode, uint32_t scheme,
                      const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *nnz_prefix, uint64_t *triplets, uint64_t cap, bool coo = false)
{
    const uint64_t esz = coo ? 10 : 24;      // bytes per entry of the output
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
    const uint32_t hw = usable_cpus();
    static const uint32_t thr_env = getenv("RC_READ_THREADS") ? (uint32_t)atoi(getenv("RC_READ_THREADS")) : 0u;   // (development: 1..16)
    const uint32_t nthr = std::max(1u, std::min<uint32_t>(std::min<uint32_t>(n, thr_env ? std::min<uint32_t>(thr_env, RC_READ_THREADS) : RC_READ_THREADS), hw));
    const int dev_now = U.device;
    // frames are claimed one at a time: the calling thread starts at once, the pool's workers join in as they wake up (their
    // wake-up, not the walk - 30 us per frame - is what a static split waited for)
    std::atomic<uint32_t> next_frame{0};
    uint64_t max_npk = 0;
    if (level == 1) for (uint32_t f = 0; f < n; ++f) max_npk = std::max<uint64_t>(max_npk, sizes[3 * f + 2]);
    auto index_range = [&](uint32_t t) {
        if (t) (void)hipSetDevice(dev_now);   // (a worker thread: page-locked memory it allocates belongs to this device's context)
        auto &BM = u.rd_bm[t]; auto &PV = u.rd_pv[t]; auto &RAW = u.rd_raw[t]; auto &all = u.rd_tmp[t]; auto &OFF = u.rd_off[t];
        BM.clear(); PV.clear(); RAW.clear(); OFF.clear();
        if (codec) {
            // what this thread is likely to collect (frames are claimed one at a time: up to three times its even share), reserved
            // in one allocation each; anything beyond still grows by doubling
            const uint64_t share = std::min<uint64_t>(n, 3ull * ((n + nthr - 1) / nthr));
            OFF.reserve(share * ((nb + TILE_BM - 1) / TILE_BM + 1));
            if (level == 1 && codec == 1) PV.reserve(share * (max_npk / 1000 + 2));   // (value-stream chunks: PIX_CHUNK = 1008 bytes each)
        }
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
            if ((r = need(6, cap * esz + 64)) != RC_OK) { (void)hipStreamSynchronize(s); return r; }
            host_async = triplets;
            triplets = reinterpret_cast<uint64_t *>(u.x[6]);
            dev_out = true;
        } else (void)hipGetLastError();
    }
    if (submit_only && !dev_out) return bail(RC_ERR_BAD_ARG, "rc_expand_frames_submit: triplets must be device or page-locked host memory");
    if (dev_out) {
        launch_expand_batch_count(d_bm, bm_stride, nb8, N, n, d_blk_cnt, d_blk_off, d_fnnz, d_fbase, s, d_pv_bytes, bit_depth, level, cap, d_err);
        launch_expand_batch_emit(d_bm, bm_stride, nb8, N, nx, n, d_blk_off, d_fbase, d_pv, pv_stride, d_pv_bytes, bit_depth, level, cap, triplets, s, d_err, coo);
    } else
        launch_expand_batch_count(d_bm, bm_stride, nb8, N, n, d_blk_cnt, d_blk_off, d_fnnz, d_fbase, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(u.h_res, d_fbase, (uint64_t)(n + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(u.h_res + n + 1, d_err, 4, hipMemcpyDeviceToHost, s));
    if (host_async && cap) HIP_TRY(hipMemcpyAsync(host_async, triplets, cap * esz, hipMemcpyDeviceToHost, s));
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
    if (coo) {
        // the three arrays keep the caller's stride (cap) on the device as in the caller's buffer: three copies of `total` entries
        if ((r = need(6, cap * esz + 64)) != RC_OK) return r;
        launch_expand_batch_emit(d_bm, bm_stride, nb8, N, nx, n, d_blk_off, d_fbase, d_pv, pv_stride, d_pv_bytes, bit_depth, level, cap, u.x[6], s, nullptr, true);
        HIP_TRY(hipGetLastError());
        uint8_t *h = reinterpret_cast<uint8_t *>(triplets);
        HIP_TRY(hipMemcpyAsync(h, u.x[6], total * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h + cap * 4, u.x[6] + cap * 4, total * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h + cap * 8, u.x[6] + cap * 8, total * 2, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return RC_OK;
    }
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

RC_EXPORT int rc_expand_frames_coo(uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t level, uint32_t op_mode, uint32_t scheme,
                                   const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *nnz_prefix, void *coo, uint64_t cap)
{
    if (level == 1 && bit_depth > 16) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_coo: values are uint16 (bit_depth <= 16)");
    return expand_run(RC_READ_SLOTS, false, nx, ny, bit_depth, level, op_mode, scheme, data, sizes, n, nnz_prefix, static_cast<uint64_t *>(coo), cap, true);
}

RC_EXPORT int rc_expand_frames_coo_submit(uint32_t slot, uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t level, uint32_t op_mode,
                                          uint32_t scheme, const uint8_t *data, const uint32_t *sizes, uint32_t n, void *coo_dev, uint64_t cap)
{
    if (slot >= RC_READ_SLOTS) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_coo_submit: slot 0 or 1");
    if (!coo_dev) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_coo_submit: the output must be device or page-locked host memory");
    if (level == 1 && bit_depth > 16) return fail(RC_ERR_BAD_ARG, "rc_expand_frames_coo: values are uint16 (bit_depth <= 16)");
    return expand_run(slot, true, nx, ny, bit_depth, level, op_mode, scheme, data, sizes, n, nullptr, static_cast<uint64_t *>(coo_dev), cap, true);
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
