// rc_zstd_dec.hip - device half of the batched stream decoders (gfx950): zstd / LZ4 blocks of many frames at once.
//
// Replaces de_compress() on the two streams of every frame the reader touches (pyrecode/recode_reader.py:393-411 ->
// recode_compressors.py:40-79) for frames inside the subset rc_zstd_dec.h describes (everything this library writes).
// The host has walked the frames' block headers and built the tables; here ONE LANE decodes ONE BLOCK - the entropy-coded
// streams of a block are serial chains (Huffman: one table step per literal; FSE: one per sequence), so blocks, not bytes,
// are the unit of parallelism, exactly as in the encoder's k_zstd_fse.  A lane regenerates its block into its own row of the
// workgroup's LDS (rows start in different banks), the wavefront then writes the rows out with coalesced stores.
#include <algorithm>

#include "rc_launch.h"
#include "rc_zstd_dec.h"

namespace rc {

__device__ __constant__ uint16_t c_ll_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                                  48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65535};
__device__ __constant__ uint8_t c_ll_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3,
                                                 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ __constant__ uint16_t c_ml_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                                  30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                                  4099, 8195, 16387, 32771, 65535};
__device__ __constant__ uint8_t c_ml_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

struct BackBits {   // backward bit reader over global memory: `bit` unread bits below the current position
    const uint8_t *p;
    int32_t bit;
    __device__ bool init(const uint8_t *src, uint32_t n)
    {
        if (n == 0 || src[n - 1] == 0) return false;
        p = src;
        bit = (int32_t)(n - 1) * 8 + (31 - __clz((int)src[n - 1]));
        return true;
    }
    // the nb <= 24 bits just below the position, without consuming (positions below 0 read as zero)
    __device__ uint32_t peek(uint32_t nb) const
    {
        const int32_t lo = bit - (int32_t)nb;
        uint32_t v = 0;
        const int32_t b0 = lo >> 3;   // may be negative
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int32_t b = b0 + i;
            const uint32_t byte = b >= 0 && b <= ((bit - 1) >> 3) ? p[b] : 0u;
            const int32_t sh = 8 * i - (lo & 7);
            if (sh >= 0) { if (sh < 32) v |= byte << sh; }
            else v |= byte >> (-sh);
        }
        return nb >= 32 ? v : (v & ((1u << nb) - 1u));
    }
    __device__ uint32_t read(uint32_t nb)
    {
        if (nb == 0) return 0;
        const uint32_t v = peek(nb);
        bit -= (int32_t)nb;
        return v;
    }
};

struct LitSource {   // the block's literals, delivered one at a time in order
    uint32_t mode;   // 0 raw, 1 RLE, 2 Huffman
    const uint8_t *raw;
    uint32_t left;
    BackBits hb;
    const uint16_t *dt;
    uint32_t log;
    __device__ uint32_t next(int *err)
    {
        if (left == 0) { *err = 1; return 0; }
        --left;
        if (mode == 0) return *raw++;
        if (mode == 1) return *raw;
        const uint32_t e = dt[hb.peek(log)];
        hb.bit -= (int32_t)(e >> 8);
        if (hb.bit < 0) *err = 1;
        return e & 0xFFu;
    }
};

// regenerate one Compressed block into `row` (cap bytes); returns the bytes produced
__device__ uint32_t zstd_block_decode(const uint8_t *c, uint32_t bs, const ZdBlock &b, const uint16_t *huf, uint32_t huf_log,
                                      const uint32_t *ll, uint32_t ll_log, const uint32_t *ml, uint32_t ml_log, uint8_t *row, uint32_t cap,
                                      int *err)
{
    const uint32_t lt = c[0] & 3u, sf = (c[0] >> 2) & 3u;
    uint32_t lhs, nlit, lit_c;
    if (lt < 2) {
        if (sf == 0 || sf == 2) { lhs = 1; nlit = c[0] >> 3; }
        else if (sf == 1) { lhs = 2; nlit = (c[0] >> 4) | ((uint32_t)c[1] << 4); }
        else { lhs = 3; nlit = (c[0] >> 4) | ((uint32_t)c[1] << 4) | ((uint32_t)c[2] << 12); }
        lit_c = lt == 0 ? nlit : 1;
    } else {
        const uint32_t lh = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
        lhs = 3; nlit = (lh >> 4) & 0x3FFu; lit_c = lh >> 14;
    }
    LitSource L;
    L.mode = lt < 2 ? lt : 2u;
    L.raw = c + lhs;
    L.left = nlit;
    L.dt = huf;
    L.log = huf_log;
    if (lt >= 2) {
        const uint32_t skip = lt == 2 ? b.tree_skip : 0u;
        if (lit_c <= skip || !L.hb.init(c + lhs + skip, lit_c - skip)) { *err = 1; return 0; }
    }
    const uint8_t *sq = c + lhs + lit_c;
    uint32_t nseq = sq[0], nsb = 1;
    if (nseq >= 128) { nseq = ((nseq - 128) << 8) + sq[1]; nsb = 2; }
    uint32_t op = 0;
    if (nseq) {
        const uint8_t *bsrc = sq + nsb + 1 + b.seq_skip;
        const uint32_t blen = (uint32_t)(c + bs - bsrc);
        BackBits fb;
        if ((int32_t)blen <= 0 || !fb.init(bsrc, blen)) { *err = 1; return 0; }
        uint32_t sl = fb.read(ll_log);   // initial states: literal length, (offset: RLE, no bits), match length
        uint32_t sm = fb.read(ml_log);
        for (uint32_t i = 0; i < nseq; ++i) {
            const uint32_t el = ll[sl], em = ml[sm];
            const uint32_t llc = el & 0xFFu, mlc = em & 0xFFu;
            if (llc > 35 || mlc > 52) { *err = 1; return op; }
            const uint32_t mlen = c_ml_base[mlc] + fb.read(c_ml_bits[mlc]);   // extra bits: (offset: none), match length, literal length
            const uint32_t llen = c_ll_base[llc] + fb.read(c_ll_bits[llc]);
            if (llen == 0 || op + llen + mlen > cap) { *err = 1; return op; }   // offset code 0 means "previous byte" only behind a literal
            for (uint32_t k = 0; k < llen; ++k) row[op++] = (uint8_t)L.next(err);
            const uint8_t prev = row[op - 1];
            if (prev == 0) op += mlen;   // the rows start out zeroed: a run of zeros (what these matches are in a bitmap) costs nothing
            else for (uint32_t k = 0; k < mlen; ++k) row[op++] = prev;
            if (i + 1 < nseq) {                                               // state updates: literal length, match length, (offset)
                sl = (el >> 16) + fb.read((el >> 8) & 0xFFu);
                sm = (em >> 16) + fb.read((em >> 8) & 0xFFu);
            }
            if (fb.bit < 0) { *err = 1; return op; }
        }
    }
    if (op + L.left > cap) { *err = 1; return op; }
    while (L.left) row[op++] = (uint8_t)L.next(err);
    return op;
}

// LZ4 block into a row (lz4_Block_format.md); the block may not reference anything in front of itself
__device__ uint32_t lz4_block_decode_row(const uint8_t *src, uint32_t n, uint8_t *row, uint32_t cap, int *err)
{
    uint32_t ip = 0, op = 0;
    while (ip < n) {
        const uint32_t token = src[ip++];
        uint32_t lit = token >> 4;
        if (lit == 15) { uint32_t x; do { if (ip >= n) { *err = 1; return op; } x = src[ip++]; lit += x; } while (x == 255); }
        if (ip + lit > n || op + lit > cap) { *err = 1; return op; }
        for (uint32_t i = 0; i < lit; ++i) row[op + i] = src[ip + i];
        ip += lit; op += lit;
        if (ip >= n) break;
        if (ip + 2 > n) { *err = 1; return op; }
        const uint32_t off = src[ip] | ((uint32_t)src[ip + 1] << 8);
        ip += 2;
        uint32_t ml = token & 15u;
        if (ml == 15) { uint32_t x; do { if (ip >= n) { *err = 1; return op; } x = src[ip++]; ml += x; } while (x == 255); }
        ml += 4;
        if (off == 0 || off > op || op + ml > cap) { *err = 1; return op; }
        for (uint32_t i = 0; i < ml; ++i) row[op + i] = row[op + i - off];
        op += ml;
    }
    return op;
}

// ROW: row capacity in bytes (multiple of 4); T: blocks (= decoding lanes) per workgroup, so that T rows + the tables fit
// the 64 KiB of static LDS.  CODEC 1: zstd Compressed blocks, 2: LZ4 compressed blocks.
// Blocks are grouped by frame (grid.y); frame_first[f] .. frame_first[f + 1] index `blocks`.
template <int ROW, int CODEC, int T>
__global__ __launch_bounds__(64) void k_block_decode(const uint8_t *__restrict__ data, const ZdBlock *__restrict__ blocks,
                                                       const uint32_t *__restrict__ frame_first, const ZdTables *__restrict__ tables,
                                                       const ZdTables *__restrict__ predef, uint8_t *__restrict__ out,
                                                       const uint64_t *__restrict__ out_base, int *__restrict__ err,
                                                       uint32_t *__restrict__ produced_out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_row[T][ROW + 4];
    __shared__ ZdTables s_t;
    __shared__ uint32_t s_pll[64], s_pml[64];
    const uint32_t f = blockIdx.y;
    const uint32_t lo = frame_first[f], hi = frame_first[f + 1];
    const uint32_t i0 = lo + blockIdx.x * T;
    if (i0 >= hi) return;
    const int lane = threadIdx.x;
    if (CODEC == 1) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables + f);
        for (uint32_t i = lane; i < sizeof(ZdTables) / 4; i += 64) reinterpret_cast<uint32_t *>(&s_t)[i] = src[i];
        s_pll[lane] = predef->ll[lane];
        s_pml[lane] = predef->ml[lane];
        __syncthreads();
    }
    for (uint32_t i = lane; i < (uint32_t)(T * (ROW + 4)) / 4; i += 64) reinterpret_cast<uint32_t *>(&s_row[0][0])[i] = 0;
    __syncthreads();
    const uint32_t bi = i0 + lane;
    ZdBlock b;
    uint32_t produced = 0;
    int e = 0;
    if (lane < T && bi < hi) {
        b = blocks[bi];
        if (b.regen > ROW) e = 1;
        else if (CODEC == 1) {
            const bool fr = b.seq_tables != 0;
            produced = zstd_block_decode(data + b.src, b.csize, b, s_t.huf, s_t.huf_log, fr ? s_t.ll : s_pll, fr ? s_t.ll_log : 6u,
                                         fr ? s_t.ml : s_pml, fr ? s_t.ml_log : 6u, s_row[lane], b.regen, &e);
        } else
            produced = lz4_block_decode_row(data + b.src, b.csize, s_row[lane], b.regen, &e);
        if (b.flex ? produced > b.regen : produced != b.regen) e = 1;
        if (b.flex && produced_out) produced_out[0] = produced;   // (at most one such block per call: a single stream's last)
        if (e) *err = 1;
    }
    __syncthreads();
    // rows -> global: lanes sweep one row at a time
    for (uint32_t r = 0; r < (uint32_t)T && i0 + r < hi; ++r) {
        const ZdBlock q = blocks[i0 + r];
        uint8_t *dst = out + out_base[f] + q.dst;
        const uint8_t *src = s_row[r];
        const uint32_t n = q.regen <= (uint32_t)ROW ? q.regen : 0u;
        if (((uintptr_t)dst & 3u) == 0) {
            for (uint32_t i = lane; i < n / 4; i += 64) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
            for (uint32_t i = (n & ~3u) + lane; i < n; i += 64) dst[i] = src[i];
        } else
            for (uint32_t i = lane; i < n; i += 64) dst[i] = src[i];
    }
}

// Raw / RLE blocks (and stored LZ4 blocks) of any size: one wavefront per 4 KiB piece
__global__ __launch_bounds__(WG) void k_block_copy(const uint8_t *__restrict__ data, const ZdBlock *__restrict__ blocks, uint32_t nblocks,
                                                     uint8_t *__restrict__ out, const uint64_t *__restrict__ out_base, uint32_t pieces_per_block)
{
    const uint32_t w = blockIdx.x * WAVES + (threadIdx.x >> 6);
    const uint32_t bi = w / pieces_per_block, piece = w % pieces_per_block;
    if (bi >= nblocks) return;
    const ZdBlock b = blocks[bi];
    const int lane = lane_id();
    uint8_t *dst = out + out_base[b.frame] + b.dst;
    const uint8_t *src = data + b.src;
    for (uint32_t o = piece * 4096u; o < b.regen; o += pieces_per_block * 4096u) {
        const uint32_t n = min(4096u, b.regen - o);
        if (b.type == 1) { const uint8_t v = src[0]; for (uint32_t i = lane; i < n; i += 64) dst[o + i] = v; }
        else for (uint32_t i = lane; i < n; i += 64) dst[o + i] = src[o + i];
    }
}

void launch_block_decode(int codec, int row, const uint8_t *data, const void *blocks, const uint32_t *frame_first, uint32_t nframes,
                         uint32_t max_blocks_per_frame, const void *tables, const void *predef, uint8_t *out, const uint64_t *out_base, int *err,
                         hipStream_t s, uint32_t *produced_out)
{
    if (!max_blocks_per_frame) return;
    const dim3 blk(64);
    const dim3 g64((max_blocks_per_frame + 63) / 64, nframes), g32((max_blocks_per_frame + 31) / 32, nframes);
    const ZdBlock *b = reinterpret_cast<const ZdBlock *>(blocks);
    const ZdTables *t = reinterpret_cast<const ZdTables *>(tables), *p = reinterpret_cast<const ZdTables *>(predef);
    if (codec == 1 && row <= 512) hipLaunchKernelGGL((k_block_decode<512, 1, 64>), g64, blk, 0, s, data, b, frame_first, t, p, out, out_base, err, produced_out);
    else if (codec == 1) hipLaunchKernelGGL((k_block_decode<1024, 1, 32>), g32, blk, 0, s, data, b, frame_first, t, p, out, out_base, err, produced_out);
    else hipLaunchKernelGGL((k_block_decode<512, 2, 64>), g64, blk, 0, s, data, b, frame_first, t, p, out, out_base, err, produced_out);
}
void launch_block_copy(const uint8_t *data, const void *blocks, uint32_t nblocks, uint32_t max_regen, uint8_t *out, const uint64_t *out_base,
                       hipStream_t s)
{
    if (!nblocks) return;
    const uint32_t ppb = std::max(1u, std::min(64u, (max_regen + 16383u) / 16384u));
    const uint32_t waves = nblocks * ppb;
    hipLaunchKernelGGL(k_block_copy, dim3((waves + WAVES - 1) / WAVES), dim3(WG), 0, s, data, reinterpret_cast<const ZdBlock *>(blocks), nblocks,
                       out, out_base, ppb);
}

size_t zd_tables_bytes() { return sizeof(ZdTables); }
size_t zd_block_bytes() { return sizeof(ZdBlock); }
void zd_predefined_tables(void *dst)
{
    static const int16_t ll[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
    static const int16_t ml[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
    ZdTables *t = reinterpret_cast<ZdTables *>(dst);
    memset(t, 0, sizeof *t);
    zd_fse_dtable(ll, 36, 6, t->ll);
    zd_fse_dtable(ml, 53, 6, t->ml);
    t->ll_log = 6;
    t->ml_log = 6;
}

}  // namespace rc
