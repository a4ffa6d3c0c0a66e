// rc_zstd.hip - Zstandard block encoding of packed binary maps on the GPU, in two halves (rc_zstd_wave.h): the byte-parallel
// half (zero runs, literals, sequence tokens) is wave-collective - fused into the reduce kernel, or k_zstd_buffer here for
// stand-alone buffers - and the FSE bitstream, a serial chain per block, runs one LANE per 512-byte block (k_zstd_fse).
// Format notes and the serial restatement used by the host-side format check: rc_zstd_block.h.
//
// Replaces `ZstdCompressor(level, write_content_size=False).compress(bitmap)` (pyrecode/recode_writer.py:175-178,
// recode_compressors.py:88).  Decoding stays with the stock library on the host (recode_compressors.py:46).
#include <algorithm>
#include <cstring>

#include <cstdlib>
#include "rc_launch.h"
#include "rc_zstd_block.h"
#include "rc_zstd_model.h"
#include "rc_zstd_wave.h"

namespace rc {

// Unfused form (seam 2 buffers, reduction level 2): wave w tokenizes block t = blockIdx.x*WAVES + w of row blockIdx.y of
// sc.bitmap (rows padded to whole blocks) exactly like the fused reduce kernel does; k_zstd_fse finishes the blocks.
__global__ __launch_bounds__(WG) void k_zstd_buffer(Scratch sc)
{
    __shared__ Lz4Lds s_lz[WAVES];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t t = blockIdx.x * WAVES + w, f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint64_t b0 = (uint64_t)t * TILE_BM;
    const uint32_t n = (uint32_t)min((uint64_t)TILE_BM, sc.nb - b0);
    const u32x2 v = reinterpret_cast<const u32x2 *>(sc.bitmap + (uint64_t)f * sc.nb_stride + b0)[lane];
    const uint64_t own = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    const bool last = t + 1 == sc.ntiles;
    uint32_t staged;
    const uint32_t word = zstd_tokenize_block(own, n, last, s_lz[w], staged);
    const uint64_t ft = (uint64_t)f * sc.ntiles + t;
    zstd_store_block(sc.blk_slots + ft * sc.blk_stride, n, last, word, staged, s_lz[w]);
    if (lane == 0) sc.blk_size[ft] = word;
}

// Second half of the fused path (rc_zstd_wave.h): thread ft turns the token list the reduce kernel left in slot ft into the
// FSE bitstream and completes the block header.  Slots that already hold a finished block (RLE / Raw) only get their size
// word cleaned.
//
// The chain itself is a few thousand cycles per block; what matters is that it never waits on memory it does not need:
// tokens arrive by 16-byte loads issued two chunks ahead, and the loop contains NO global store (gfx9 counts loads and
// stores on one in-order counter, so a store inside the loop would put its whole write latency in front of the next token).
// The bitstream is collected in a per-lane LDS row and written out afterwards.  A lane whose bitstream outgrows its row
// (dense blocks) re-runs the chain with stores straight into the slot, in place: a sequence costs at most 29 bits and its
// token is 32, so the write position never passes the read position.
#ifndef RC_FSE_ROW
#define RC_FSE_ROW 48
#endif
#ifndef RC_FSE_T
#define RC_FSE_T 128
#endif
constexpr int FSE_ROW = RC_FSE_ROW;  // dwords of bitstream kept in LDS per lane (192 bytes; a 1 %-sparsity block needs about 15)
// 128 threads = 25 KB of LDS per workgroup: in pipelined mode this kernel is dispatched while the next batch's reduce kernel
// fills the CUs (3 workgroups x 39.5 KB of the 160 KB), and a 50 KB workgroup would have to wait for one of them to leave
constexpr int FSE_T = RC_FSE_T;

// fitted != 0: the tables are a ctx's fitted ones and the blocks' sequences are in Repeat_Mode (ZW_SEQ is reported in the
// size word, next to the tokenizer's ZW_TREE, for k_scan_frames to place the frame's definitions).
__global__ __launch_bounds__(FSE_T) void k_zstd_fse(Scratch sc, uint32_t nslots, const ZstdTables *__restrict__ tables, uint32_t fitted)
{
    __shared__ ZstdTables T;
    __shared__ uint32_t s_row[FSE_T][FSE_ROW + 1];  // + 1: rows start in different banks
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&T);
        for (uint32_t i = threadIdx.x; i < sizeof(ZstdTables) / 4; i += FSE_T) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t ft = blockIdx.x * FSE_T + threadIdx.x;
    if (ft >= nslots) return;
    const uint32_t word = sc.blk_size[ft];
    if (word & ZW_FINAL) { sc.blk_size[ft] = word & (0xFFFFu | ZW_TREE); return; }
    const uint32_t P = word & 0xFFFFu, nseq = (word >> 16) & 0xFFu;
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(sc.blk_slots + (uint64_t)ft * sc.blk_stride);
    const ZW4 *tok4 = reinterpret_cast<const ZW4 *>(slot32 + (zstd_token_offset(P) >> 2));
    const uint32_t w0 = P >> 2, nb0 = 8 * (P & 3u);
    const uint64_t acc0 = nb0 ? (uint64_t)(slot32[w0] & ((1u << nb0) - 1u)) : 0ull;
    uint32_t *row = s_row[threadIdx.x];
    uint32_t o = 0;
    uint64_t acc = acc0;
    uint32_t nb = fse_chain(tok4, nseq, acc, nb0, T, [&](uint32_t v) { if (o < (uint32_t)FSE_ROW) row[o] = v; ++o; });
    // at most 50 bits are left: one or two more dwords
    uint32_t ndw = o + ((nb + 31) >> 5);
    const uint32_t end = 4 * (w0 + o) + ((nb + 7) >> 3);  // first byte behind the bitstream
    if (ndw <= (uint32_t)FSE_ROW) {
        if (nb) { row[o] = (uint32_t)acc; if (nb > 32) row[o + 1] = (uint32_t)(acc >> 32); }
        for (uint32_t i = 0; i < ndw; ++i) slot32[w0 + i] = row[i];
    } else {  // does not fit the LDS row: again, storing in place
        o = 0;
        acc = acc0;
        nb = fse_chain(tok4, nseq, acc, nb0, T, [&](uint32_t v) { slot32[w0 + o] = v; ++o; });
        if (nb) { slot32[w0 + o] = (uint32_t)acc; if (nb > 32) slot32[w0 + o + 1] = (uint32_t)(acc >> 32); }
    }
    const uint32_t content = end - 3;
    const uint32_t lastbit = (ft % sc.ntiles) + 1 == sc.ntiles ? 1u : 0u;
    slot32[0] |= lastbit | (2u << 1) | (content << 3);  // bytes 0..2 were left zero; byte 3 is the literals header
    sc.blk_size[ft] = end | (word & ZW_TREE) | (fitted ? ZW_SEQ : 0u);
}
void launch_zstd_fse(const Scratch &sc, uint32_t B, const void *tables_dev, bool fitted, hipStream_t s)
{
    const uint32_t nslots = B * sc.ntiles;
    hipLaunchKernelGGL(k_zstd_fse, dim3((nslots + FSE_T - 1) / FSE_T), dim3(FSE_T), 0, s, sc, nslots,
                       reinterpret_cast<const ZstdTables *>(tables_dev), fitted ? 1u : 0u);
}

void launch_zstd_tokenize_rows(const Scratch &sc, uint32_t B, hipStream_t s)
{
    hipLaunchKernelGGL(k_zstd_buffer, dim3((sc.ntiles + WAVES - 1) / WAVES, B), dim3(WG), 0, s, sc);
}

void launch_zstd_encode_blocks(const Scratch &sc, uint32_t B, const void *tables_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_zstd_buffer, dim3((sc.ntiles + WAVES - 1) / WAVES, B), dim3(WG), 0, s, sc);
    launch_zstd_fse(sc, B, tables_dev, false, s);
}

// ---- sample for the modelled encoder ----------------------------------------------------------------------------------------
// Runs over the slots the PLAIN tokenizer left (before k_zstd_fse): literal bytes, literal-length / match-length codes of the
// tokens, and the bytes of the tiles' packed residual streams, into the histograms rc_zstd_model.h fits its tables to.
__global__ __launch_bounds__(WG) void k_zstd_sample(Scratch sc, uint32_t B, uint32_t with_pix, uint32_t depth, ZstdSample *__restrict__ h)
{
    // histograms per workgroup in LDS, added to the global ones once at the end (a few hundred counters shared by every tile
    // of the sample: global atomics alone took 94 ms on an 11520x8184 frame)
    __shared__ ZstdSample s_h;
    constexpr uint32_t NW = sizeof(ZstdSample) / 4;
    for (uint32_t i = threadIdx.x; i < NW; i += WG) reinterpret_cast<uint32_t *>(&s_h)[i] = 0;
    __syncthreads();
    const uint32_t f = blockIdx.y;
    const int lane = lane_id();
    for (uint32_t t = blockIdx.x * WAVES + (threadIdx.x >> 6); t < sc.ntiles && f < B; t += gridDim.x * WAVES) {
        const uint64_t ft = (uint64_t)f * sc.ntiles + t;
        const uint32_t word = sc.blk_size[ft];
        if (!(word & ZW_FINAL)) {
            const uint8_t *slot = sc.blk_slots + ft * sc.blk_stride;
            const uint32_t P = word & 0xFFFFu, nseq = word >> 16;
            const uint32_t b3 = slot[3];
            const uint32_t lh = (b3 & 4u) ? 2u : 1u;
            const uint32_t nlit = lh == 1 ? (b3 >> 3) : ((b3 >> 4) | ((uint32_t)slot[4] << 4));
            for (uint32_t i = lane; i < nlit; i += 64) atomicAdd(&s_h.lit[slot[3 + lh + i]], 1u);
            const uint32_t *tok = reinterpret_cast<const uint32_t *>(slot + zstd_token_offset(P));
            for (uint32_t i = lane; i < nseq; i += 64) {
                const uint32_t k = tok[i];
                atomicAdd(&s_h.ll[k & 63u], 1u);
                atomicAdd(&s_h.ml[(k >> 6) & 63u], 1u);
            }
        }
        if ((word & 0xFFFFu) != 4u || !(word & ZW_FINAL)) {   // not an all-zero block (those are RLE blocks in every form): all its bytes
            const uint64_t b0 = (uint64_t)t * TILE_BM;
            const uint32_t nbm = (uint32_t)(sc.nb - b0 < (uint64_t)TILE_BM ? sc.nb - b0 : (uint64_t)TILE_BM);
            const uint8_t *bm = sc.bitmap + (uint64_t)f * sc.nb_stride + b0;
            for (uint32_t i = lane; i < nbm; i += 64) atomicAdd(&s_h.all[bm[i]], 1u);
            if (lane == 0) atomicAdd(&s_h.nblk, 1u);
        }
        if (with_pix) {
            const uint32_t nbytes = (sc.tile_cnt[ft] * depth + 7) >> 3;
            const uint8_t *p = residual_src(sc, ft, 0u, sc.tile_cnt[ft], depth);   // (zstd: the combined slots' fixed offset, no block size needed)
            for (uint32_t i = lane; i < nbytes; i += 64) atomicAdd(&s_h.pix[p[i]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NW; i += WG) {
        const uint32_t v = reinterpret_cast<uint32_t *>(&s_h)[i];
        if (v) atomicAdd(reinterpret_cast<uint32_t *>(h) + i, v);
    }
}
void launch_zstd_sample(const Scratch &sc, uint32_t B, bool with_pix, uint32_t depth, void *sample_dev, hipStream_t s)
{
    const uint32_t gx = std::min<uint32_t>((sc.ntiles + WAVES - 1) / WAVES, 512u);   // workgroups loop over the tiles of their frame
    hipLaunchKernelGGL(k_zstd_sample, dim3(gx, B), dim3(WG), 0, s, sc, B, with_pix ? 1u : 0u, depth,
                       reinterpret_cast<ZstdSample *>(sample_dev));
}
size_t zstd_model_bytes() { return sizeof(ZstdModel); }
size_t zstd_sample_bytes() { return sizeof(ZstdSample); }
void zstd_model_from_sample(const void *sample_host, void *model_host, uint32_t speed_permille)
{
    ZstdSample h = *reinterpret_cast<const ZstdSample *>(sample_host);
    if (getenv("RC_ZSTD_SEQ_ALWAYS")) h.nblk = 0;   // (A/B runs: never the literals-only block form)
    if (getenv("RC_ZSTD_LITS_ALWAYS")) { for (auto &v : h.ll) v = 0x100000; h.nblk = h.nblk ? h.nblk : 1; }   // (A/B runs: always; the sequences priced out)
    zm_build_model(h, *reinterpret_cast<ZstdModel *>(model_host), speed_permille);
}

size_t zstd_tables_bytes() { return sizeof(ZstdTables); }
void zstd_tables_host(void *dst)
{
    ZstdTables t;
    zstd_build_tables(t);
    memcpy(dst, &t, sizeof t);
}

// Stand-alone frame of an arbitrary buffer (seam 2): header + the encoded blocks, nothing behind them (the final block
// carries Last_Block).  One wavefront per block copy.
__global__ __launch_bounds__(WG) void k_zstd_gather(Scratch sc, uint8_t *__restrict__ out)
{
    const uint32_t t = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (t >= sc.ntiles) return;
    if (t == 0 && lane_id() == 0) {
        out[0] = 0x28; out[1] = 0xB5; out[2] = 0x2F; out[3] = 0xFD;
        out[4] = 0x00;  // no content size, window descriptor follows
        out[5] = 0x00;  // 1 KiB window (blocks are 512 bytes, offsets are 1)
    }
    const uint8_t *src = sc.blk_slots + (uint64_t)t * BLK_SLOT;
    uint8_t *dst = out + 6 + sc.blk_off[t];
    const uint32_t n = sc.blk_size[t];
    for (uint32_t i = lane_id(); i < n; i += 64) dst[i] = src[i];
}
void launch_zstd_gather(const Scratch &sc, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_zstd_gather, dim3((sc.ntiles + WAVES - 1) / WAVES), dim3(WG), 0, s, sc, out);
}

}  // namespace rc
