// rc_zstd_dec.hip - device half of the batched stream decoders (gfx950): zstd / LZ4 blocks of many frames at once.
//
// Replaces de_compress() on the two streams of every frame the reader touches (pyrecode/recode_reader.py:393-411 ->
// recode_compressors.py:40-79) for frames inside the subset rc_zstd_dec.h describes (everything this library writes).
// The host has walked the frames' block headers and built the tables; here ONE LANE decodes ONE BLOCK - the entropy-coded
// streams of a block are serial chains (Huffman: one table step per literal; FSE: one per sequence), so blocks, not bytes,
// are the unit of parallelism, exactly as in the encoder's k_zstd_fse.  A workgroup stages the compressed bytes of its blocks in
// LDS; a lane regenerates its block straight into (zeroed) global memory, storing only what is not zero (k_block_decode).
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

// Backward bit reader.  `p` may point into LDS or global memory (generic address; the kernels instantiate the decoder once per
// address space).  EVERY read fetches the two aligned dwords around the position - no state, no branch: in a wavefront the lanes
// cross dword boundaries at different symbols, so a reader that keeps dwords in registers and refills on demand executes its refill
// path at almost every step anyway, under nested exec masks (measured: ~1000 cycles per Huffman symbol; this form: see DESIGN.md).
// Bits below the stream's first are whatever lies there: a valid stream never depends on them (a Huffman step is decided by the
// code's own bits, an FSE read never crosses the start), a corrupt one runs into `bit < 0`, which the callers check.
struct BackBits {
    const uint32_t *pa;   // the stream's first byte, rounded down to a dword
    int32_t bit;          // unread bits, counted from bit 0 of pa[0]
    int32_t first;        // the stream's first bit in the same count (8 * misalignment)
    __device__ bool init(const uint8_t *src, uint32_t n)
    {
        if (n == 0) return false;
        const uint32_t lastb = src[n - 1];
        if (lastb == 0) return false;
        const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
        pa = reinterpret_cast<const uint32_t *>(src - mis);
        first = 8 * (int32_t)mis;
        bit = first + (int32_t)(n - 1) * 8 + (31 - __clz((int)lastb));
        return true;
    }
    __device__ int32_t left() const { return bit - first; }   // < 0: the stream was over-read
    // the nb <= 24 bits just below the position, without consuming
    __device__ uint32_t peek(uint32_t nb) const
    {
        const int32_t i = max(((bit - 1) >> 5) - 1, 0);         // window = dwords i, i + 1: bits [32 i, 32 i + 64) hold [bit - nb, bit)
        const uint32_t lo = pa[i], hi = pa[i + 1];
        // fewer than nb bits left above dword 0 (a Huffman peek at a stream's very start): the bits that exist stay at the field's top
        const int32_t s = bit - (int32_t)nb - 32 * i;
        const uint64_t w = (((uint64_t)hi << 32) | lo) << (uint32_t)max(-s, 0);
        return (uint32_t)(w >> (uint32_t)max(s, 0)) & ((1u << nb) - 1u);
    }
    __device__ uint32_t read(uint32_t nb)
    {
        const uint32_t v = peek(nb);   // (nb == 0: mask 0)
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
        if (hb.left() < 0) *err = 1;
        return e & 0xFFu;
    }
};

// Where a lane's regenerated bytes go: straight to the block's place in global memory, which the caller has ZEROED.
//   SPARSE (binary maps): only non-zero bytes are stored - a zero literal or a run of zeros costs nothing;
//   dense (value streams): bytes are collected eight at a time and leave as one 8-byte store when the destination is 8-byte aligned.
template <bool SPARSE>
struct Sink {
    uint8_t *g;
    uint32_t op;
    uint64_t acc;
    bool al;
    __device__ void init(uint8_t *dst) { g = dst; op = 0; acc = 0; al = ((uintptr_t)dst & 7u) == 0; }
    __device__ void put(uint32_t v)
    {
        if (SPARSE) { if (v) g[op] = (uint8_t)v; ++op; return; }
        acc |= (uint64_t)v << (8 * (op & 7u));
        ++op;
        if ((op & 7u) == 0) flush8();
    }
    __device__ void flush8()   // the eight bytes in front of op
    {
        if (al) *reinterpret_cast<uint64_t *>(g + op - 8) = acc;
        else for (int j = 0; j < 8; ++j) g[op - 8 + j] = (uint8_t)(acc >> (8 * j));
        acc = 0;
    }
    __device__ void fill(uint32_t v, uint32_t n)   // n copies of v
    {
        if (SPARSE) { if (v) for (uint32_t k = 0; k < n; ++k) g[op + k] = (uint8_t)v; op += n; return; }
        for (uint32_t k = 0; k < n; ++k) put(v);
    }
    __device__ void finish()
    {
        if (SPARSE) return;
        const uint32_t k = op & 7u;
        for (uint32_t j = 0; j < k; ++j) g[op - k + j] = (uint8_t)(acc >> (8 * j));
    }
};

// regenerate one Compressed block (content c, bs bytes; generic address) into the sink, at most cap bytes; returns the bytes produced
// b.seq_tables == 2 (compact lists): the block's own modes byte decides - predefined tables (llm 0, one RLE offset byte behind the
// modes), the frame's tables defined right here (llm 2: b.seq_skip bytes of descriptions) or repeated (llm 3); pll / pml then
// hold the predefined tables and ll / ml the frame's.
template <bool SPARSE>
__device__ uint32_t zstd_block_decode(const uint8_t *c, uint32_t bs, const ZdBlock &b, const uint16_t *huf, uint32_t huf_log,
                                      const uint32_t *ll, uint32_t ll_log, const uint32_t *ml, uint32_t ml_log, const uint32_t *llx,
                                      const uint32_t *mlx, Sink<SPARSE> &o, uint32_t cap, int *err, const uint32_t *pll = nullptr,
                                      const uint32_t *pml = nullptr)
{
    const uint32_t c0 = c[0];
    const uint32_t lt = c0 & 3u, sf = (c0 >> 2) & 3u;
    uint32_t lhs, nlit, lit_c;
    if (lt < 2) {
        if (sf == 0 || sf == 2) { lhs = 1; nlit = c0 >> 3; }
        else if (sf == 1) { lhs = 2; nlit = (c0 >> 4) | ((uint32_t)c[1] << 4); }
        else { lhs = 3; nlit = (c0 >> 4) | ((uint32_t)c[1] << 4) | ((uint32_t)c[2] << 12); }
        lit_c = lt == 0 ? nlit : 1;
    } else {
        const uint32_t lh = c0 | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
        lhs = 3; nlit = (lh >> 4) & 0x3FFu; lit_c = lh >> 14;
    }
    if (lhs + lit_c + 1 > bs) { *err = 1; return 0; }
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
    uint32_t prev = 0;
    if (nseq) {
        uint32_t seq_skip = b.seq_skip;
        if (b.seq_tables == 2) {
            const uint32_t llm = sq[nsb] >> 6;
            if (llm == 0) { ll = pll; ml = pml; ll_log = ml_log = 6; seq_skip = 1; }
            else if (llm == 3) seq_skip = 0;
        }
        const uint8_t *bsrc = sq + nsb + 1 + seq_skip;
        const uint32_t blen = (uint32_t)(c + bs - bsrc);
        BackBits fb;
        if ((int32_t)blen <= 0 || !fb.init(bsrc, blen)) { *err = 1; return 0; }
        uint32_t sl = fb.read(ll_log);   // initial states: literal length, (offset: RLE, no bits), match length
        uint32_t sm = fb.read(ml_log);
        for (uint32_t i = 0; i < nseq; ++i) {
            const uint32_t el = ll[sl], em = ml[sm];
            const uint32_t llc = el & 0xFFu, mlc = em & 0xFFu;
            if (llc > 35 || mlc > 52) { *err = 1; return o.op; }
            const uint32_t mx = mlx[mlc], lx = llx[llc];                      // base | extra bits << 16
            const uint32_t mlen = (mx & 0xFFFFu) + fb.read(mx >> 16);         // extra bits: (offset: none), match length, literal length
            const uint32_t llen = (lx & 0xFFFFu) + fb.read(lx >> 16);
            if (llen == 0 || o.op + llen + mlen > cap) { *err = 1; return o.op; }   // offset code 0 means "previous byte" only behind a literal
            if (L.mode == 2) {   // Huffman-coded literals: the bare table step (the lanes wait for each other in this loop: keep it short)
                if (L.left < llen) { *err = 1; return o.op; }
                for (uint32_t k = 0; k < llen; ++k) {
                    const uint32_t e = L.dt[L.hb.peek(L.log)];
                    L.hb.bit -= (int32_t)(e >> 8);
                    prev = e & 0xFFu;
                    o.put(prev);
                }
                L.left -= llen;
                if (L.hb.left() < 0) { *err = 1; return o.op; }
            } else
                for (uint32_t k = 0; k < llen; ++k) { prev = L.next(err); o.put(prev); }
            o.fill(prev, mlen);   // (a run of zeros - what these matches are in a binary map - costs nothing in the sparse sink)
            if (i + 1 < nseq) {                                               // state updates: literal length, match length, (offset)
                sl = (el >> 16) + fb.read((el >> 8) & 0xFFu);
                sm = (em >> 16) + fb.read((em >> 8) & 0xFFu);
            }
            if (fb.left() < 0) { *err = 1; return o.op; }
        }
    }
    if (o.op + L.left > cap) { *err = 1; return o.op; }
    {
        // Huffman-coded literals, eight at a time (the body of a value-stream chunk: 1008 of them and nothing else - and, since round 4, of a
        // binary-map block of the literals-only form: all its 512 bytes).  While at least 8 * log bits are left no step of a group can
        // over-read, so the group needs no checks; its eight bytes leave as one store (the sparse sink: only when one of them is not zero -
        // the output was zeroed).
        if (L.mode == 2 && o.al && (o.op & 7u) == 0) {
            const uint32_t log = L.log;
            while (L.left >= 8 && L.hb.left() >= 8 * (int32_t)log) {
                uint64_t acc = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t e = L.dt[L.hb.peek(log)];
                    L.hb.bit -= (int32_t)(e >> 8);
                    acc |= (uint64_t)(e & 0xFFu) << (8 * j);
                }
                if (!SPARSE || acc) *reinterpret_cast<uint64_t *>(o.g + o.op) = acc;
                o.op += 8;
                L.left -= 8;
            }
        }
    }
    while (L.left) o.put(L.next(err));
    o.finish();
    return o.op;
}

// LZ4 block (lz4_Block_format.md) into the sparse sink; the block may not reference anything in front of itself.  A match is
// read back from the lane's own earlier output (bytes it skipped read as the zeros the caller put there), EIGHT bytes per load:
// the event parser's matches (rc_lz4_block.h) sit at real offsets, and byte-wise read-back made every copied byte a dependent
// round trip to L2 (0.87 ms for the 262 144 blocks of 64 binary maps; this form: see DESIGN.md).
__device__ uint32_t lz4_block_decode(const uint8_t *src, uint32_t n, Sink<true> &o, uint32_t cap, int *err)
{
    uint32_t ip = 0;
    uint32_t prev = 0;   // the byte in front of o.op
    while (ip < n) {
        const uint32_t token = src[ip++];
        uint32_t lit = token >> 4;
        if (lit == 15) { uint32_t x; do { if (ip >= n) { *err = 1; return o.op; } x = src[ip++]; lit += x; } while (x == 255); }
        if (ip + lit > n || o.op + lit > cap) { *err = 1; return o.op; }
        for (uint32_t i = 0; i < lit; ++i) { prev = src[ip + i]; o.put(prev); }
        ip += lit;
        if (ip >= n) break;
        if (ip + 2 > n) { *err = 1; return o.op; }
        const uint32_t off = src[ip] | ((uint32_t)src[ip + 1] << 8);
        ip += 2;
        uint32_t ml = token & 15u;
        if (ml == 15) { uint32_t x; do { if (ip >= n) { *err = 1; return o.op; } x = src[ip++]; ml += x; } while (x == 255); }
        ml += 4;
        if (off == 0 || off > o.op || o.op + ml > cap) { *err = 1; return o.op; }
        if (off == 1) o.fill(prev, ml);   // the byte in front, repeated: known without reading anything back
        else {
            for (uint32_t i = 0; i < ml;) {
                const uint32_t k = min(min(8u, ml - i), off);   // (off < 8: the pattern repeats - only bytes already written are taken)
                uint64_t v;
                __builtin_memcpy(&v, o.g + o.op - off, 8);       // any alignment; may run past o.op, inside the (padded) output
                v &= k == 8 ? ~0ull : ((1ull << (8 * k)) - 1ull);
                prev = (uint32_t)(v >> (8 * (k - 1))) & 0xFFu;
                for (uint64_t q = v; q;) {                       // store what is not zero
                    const uint32_t j = (uint32_t)__builtin_ctzll(q) >> 3;
                    o.g[o.op + j] = (uint8_t)(v >> (8 * j));
                    q &= ~(0xFFull << (8 * j));
                }
                o.op += k;
                i += k;
            }
        }
    }
    return o.op;
}

// ONE LANE decodes ONE BLOCK, T blocks of one frame (grid.y) per workgroup; lists[f] = the frame's block list (which may lie in
// page-locked HOST memory: every entry is read once).
// The blocks of a workgroup are neighbours in the stored stream: their compressed bytes - one contiguous span, the 3-byte headers
// in between included - are first copied into LDS with coalesced 16-byte loads, and every lane reads ITS block from there (a lane
// whose block does not lie inside the staged span reads global memory instead: same code, generic addresses).  This is what
// makes the kernel fast: a block's decoding is a chain of some hundred dependent reads of its compressed bytes, and from global
// memory each of them is a full memory latency.  (Before: rows of regenerated bytes in LDS, compressed bytes from global memory -
// 1.18 ms for the 262 144 binary-map blocks of 64 frames of 4096 x 4096.)  The regenerated bytes go straight to global memory
// (Sink), which the caller has zeroed.
// CODEC 1: zstd Compressed blocks, 2: LZ4 compressed blocks.  SPAN: staged bytes (multiple of 16).
template <int CODEC, bool SPARSE, int T, int SPAN>
__global__ __launch_bounds__(T) void k_block_decode(const uint8_t *__restrict__ data, const ZdFrameList *__restrict__ lists,
                                                      const ZdTables *__restrict__ tables,
                                                      const ZdTables *__restrict__ predef, uint8_t *__restrict__ out,
                                                      const uint64_t *__restrict__ out_base, int *__restrict__ err,
                                                      uint32_t *__restrict__ produced_out)
{
    __shared__ u32x4 s_span[SPAN / 16];
    __shared__ ZdTables s_t;
    __shared__ uint32_t s_pll[64], s_pml[64];
    __shared__ uint32_t s_llx[36], s_mlx[53];   // code -> base | extra bits << 16 (from constant memory each lookup was a vector load)
    const uint32_t f = blockIdx.y;
    const ZdBlock *__restrict__ blocks = lists[f].p;
    const uint32_t hi = lists[f].n;
    const uint32_t i0 = blockIdx.x * T;
    if (i0 >= hi) return;
    const uint32_t tid = threadIdx.x;
    if (CODEC == 1) {
        if (tid < 36) s_llx[tid] = c_ll_base[tid] | ((uint32_t)c_ll_bits[tid] << 16);
        if (tid < 53) s_mlx[tid] = c_ml_base[tid] | ((uint32_t)c_ml_bits[tid] << 16);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables + f);
        for (uint32_t i = tid; i < sizeof(ZdTables) / 4; i += T) reinterpret_cast<uint32_t *>(&s_t)[i] = src[i];
        if (tid < 64) { s_pll[tid] = predef->ll[tid]; s_pml[tid] = predef->ml[tid]; }
    }
    const uint32_t ilast = min(i0 + (uint32_t)T, hi) - 1;
    const uint64_t span0 = blocks[i0].src & ~15ull;
    const uint64_t span_end = blocks[ilast].src + blocks[ilast].csize;
    const uint32_t staged = span_end > span0 ? (uint32_t)min<uint64_t>((span_end - span0 + 15) & ~15ull, (uint64_t)SPAN) : 0u;
    {
        const u32x4 *g = reinterpret_cast<const u32x4 *>(data + span0);   // (data is an allocation's start: 16-byte aligned)
        for (uint32_t i = tid; i < staged / 16; i += T) s_span[i] = g[i];
    }
    __syncthreads();
    const uint32_t bi = i0 + tid;
    if (bi >= hi) return;
    const ZdBlock b = blocks[bi];
    int e = 0;
    const bool in_lds = b.src >= span0 && b.src - span0 + b.csize <= staged;
    Sink<SPARSE> o;
    o.init(out + out_base[f] + b.dst);
    uint32_t produced = 0;
    // Two copies of the decoder, one per address space of the compressed bytes: with ONE generic pointer every read would be a
    // flat load, which is ordered behind the lane's outstanding global stores (flat and global memory instructions share vmcnt) -
    // a full store latency per few symbols (measured: 300 ns per literal of a value-stream chunk).
    auto run = [&](const uint8_t *c) {
        if (CODEC == 1) {
            const bool fr = b.seq_tables != 0;
            return zstd_block_decode<SPARSE>(c, b.csize, b, s_t.huf, s_t.huf_log, fr ? s_t.ll : s_pll, fr ? s_t.ll_log : 6u, fr ? s_t.ml : s_pml,
                                             fr ? s_t.ml_log : 6u, s_llx, s_mlx, o, b.regen, &e);
        } else {
            if constexpr (SPARSE) return lz4_block_decode(c, b.csize, o, b.regen, &e);
            else return 0u;
        }
    };
    if (b.csize == 0) e = 1;
    else if (in_lds) produced = run(reinterpret_cast<const uint8_t *>(s_span) + (uint32_t)(b.src - span0));
    else produced = run(data + b.src);
    if (b.flex ? produced > b.regen : produced != b.regen) e = 1;
    if (b.flex && produced_out) produced_out[0] = produced;   // (at most one such block per call: a single stream's last)
    if (e) *err = 1;
}

// The binary-map streams of frames THIS library's shape of encoder wrote ("uniform": every block regenerates TILE_BM bytes, the
// last one the rest; rc_reader.hip checks that on the host) need no 32-byte entry per block: the host walk leaves ONE dword per block,
// the offset of its header inside the frame's stream (n + 1 of them: the last is the stream's end), and the lane reads type, size and
// table modes from the block itself.  The lists are read over the link (page-locked host memory, once): 1 MB instead of 8 MB per 64
// frames of 4096 x 4096 - with full entries the link, not the decoder, set this kernel's time (315 us; entries in device memory: 159).
//   lists[f].p   const uint32_t *: header offsets relative to src_base[f];  lists[f].pad = tree_skip | seq_skip << 8
// CODEC 1: zstd (3-byte block headers; Raw and RLE blocks are handled in line), 2: LZ4 frame blocks (4-byte size words, bit 31 = stored).
template <int CODEC, int T, int SPAN>
__global__ __launch_bounds__(T) void k_bitmap_decode_c(const uint8_t *__restrict__ data, const ZdFrameList *__restrict__ lists,
                                                         const uint64_t *__restrict__ src_base, const ZdTables *__restrict__ tables,
                                                         const ZdTables *__restrict__ predef, uint8_t *__restrict__ out,
                                                         const uint64_t *__restrict__ out_base, uint64_t nb, int *__restrict__ err)
{
    __shared__ u32x4 s_span[SPAN / 16];
    __shared__ ZdTables s_t;
    __shared__ uint32_t s_pll[64], s_pml[64];
    __shared__ uint32_t s_llx[36], s_mlx[53];
    const uint32_t f = blockIdx.y;
    const uint32_t *__restrict__ offs = reinterpret_cast<const uint32_t *>(lists[f].p);
    const uint32_t hi = lists[f].n, skips = lists[f].pad;
    const uint32_t i0 = blockIdx.x * T;
    if (i0 >= hi) return;
    const uint32_t tid = threadIdx.x;
    if (CODEC == 1) {
        if (tid < 36) s_llx[tid] = c_ll_base[tid] | ((uint32_t)c_ll_bits[tid] << 16);
        if (tid < 53) s_mlx[tid] = c_ml_base[tid] | ((uint32_t)c_ml_bits[tid] << 16);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables + f);
        for (uint32_t i = tid; i < sizeof(ZdTables) / 4; i += T) reinterpret_cast<uint32_t *>(&s_t)[i] = src[i];
        if (tid < 64) { s_pll[tid] = predef->ll[tid]; s_pml[tid] = predef->ml[tid]; }
    }
    const uint64_t fb = src_base[f];
    const uint32_t iend = min(i0 + (uint32_t)T, hi);
    const uint64_t span0 = (fb + offs[i0]) & ~15ull;
    const uint64_t span_end = fb + offs[iend];
    const uint32_t staged = (uint32_t)min<uint64_t>((span_end - span0 + 15) & ~15ull, (uint64_t)SPAN);
    {
        const u32x4 *g = reinterpret_cast<const u32x4 *>(data + span0);
        for (uint32_t i = tid; i < staged / 16; i += T) s_span[i] = g[i];
    }
    __syncthreads();
    const uint32_t bi = i0 + tid;
    if (bi >= hi) return;
    const uint64_t h0 = fb + offs[bi], h1 = fb + offs[bi + 1];
    constexpr uint32_t HDR = CODEC == 1 ? 3u : 4u;
    int e = 0;
    if (h1 < h0 + HDR) e = 1;
    const uint32_t regen = (uint32_t)min<uint64_t>((uint64_t)TILE_BM, nb - (uint64_t)bi * TILE_BM);
    Sink<true> o;
    o.init(out + out_base[f] + (uint64_t)bi * TILE_BM);
    const bool in_lds = h0 >= span0 && h1 - span0 <= staged;
    auto run = [&](const uint8_t *h) -> uint32_t {   // h: the block's header (LDS or global: one instantiation each)
        ZdBlock b;
        b.src = h0 + HDR; b.csize = (uint32_t)(h1 - h0) - HDR; b.dst = 0; b.regen = regen; b.frame = f; b.flex = 0;
        b.tree_skip = (uint8_t)skips; b.seq_skip = (uint8_t)(skips >> 8); b.seq_tables = 2;
        const uint8_t *c = h + HDR;
        if (CODEC == 1) {
            const uint32_t hd = (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16);
            const uint32_t type = (hd >> 1) & 3u, bs = hd >> 3;
            if (type == 2) {
                if (bs != b.csize) { e = 1; return 0; }
                return zstd_block_decode<true>(c, b.csize, b, s_t.huf, s_t.huf_log, s_t.ll, s_t.ll_log, s_t.ml, s_t.ml_log, s_llx, s_mlx, o, regen, &e,
                                               s_pll, s_pml);
            }
            if (bs != regen || b.csize != (type == 1 ? 1u : bs)) { e = 1; return 0; }
            if (type == 1) { o.fill(c[0], regen); return regen; }
            for (uint32_t k = 0; k < regen; ++k) o.put(c[k]);
            return regen;
        } else {
            const uint32_t wd = (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16) | ((uint32_t)h[3] << 24);
            if ((wd & 0x7FFFFFFFu) != b.csize) { e = 1; return 0; }
            if (wd >> 31) {
                if (b.csize != regen) { e = 1; return 0; }
                for (uint32_t k = 0; k < regen; ++k) o.put(c[k]);
                return regen;
            }
            return lz4_block_decode(c, b.csize, o, regen, &e);
        }
    };
    uint32_t produced = 0;
    if (!e) produced = in_lds ? run(reinterpret_cast<const uint8_t *>(s_span) + (uint32_t)(h0 - span0)) : run(data + h0);
    if (produced != regen) e = 1;
    if (e) *err = 1;
}

// Raw / RLE blocks (and stored LZ4 blocks) of any size: one wavefront per 4 KiB piece
__global__ __launch_bounds__(WG) void k_block_copy(const uint8_t *__restrict__ data, const ZdFrameList *__restrict__ lists, uint32_t nlists,
                                                     uint8_t *__restrict__ out, const uint64_t *__restrict__ out_base, uint32_t pieces_per_block)
{
    const uint32_t w = blockIdx.x * WAVES + (threadIdx.x >> 6);
    uint32_t bi = w / pieces_per_block;
    const uint32_t piece = w % pieces_per_block;
    uint32_t l = 0;
    while (l < nlists && bi >= lists[l].n) { bi -= lists[l].n; ++l; }   // (a handful of lists: one per indexing thread)
    if (l >= nlists) return;
    const ZdBlock b = lists[l].p[bi];
    const int lane = lane_id();
    uint8_t *dst = out + out_base[b.frame] + b.dst;
    const uint8_t *src = data + b.src;
    for (uint32_t o = piece * 4096u; o < b.regen; o += pieces_per_block * 4096u) {
        const uint32_t n = min(4096u, b.regen - o);
        if (b.type == 1) { const uint8_t v = src[0]; for (uint32_t i = lane; i < n; i += 64) dst[o + i] = v; }
        else for (uint32_t i = lane; i < n; i += 64) dst[o + i] = src[o + i];
    }
}

// row: the largest number of bytes a block regenerates (<= 512: the blocks of a binary map; more: the chunks of a value stream,
// few and large - fewer lanes per workgroup so that their compressed bytes still fit the LDS span).  `out` must be zeroed.
void launch_block_decode(int codec, int row, const uint8_t *data, const void *frame_lists, uint32_t nframes, uint32_t max_blocks_per_frame,
                         const void *tables, const void *predef, uint8_t *out, const uint64_t *out_base, int *err, hipStream_t s,
                         uint32_t *produced_out)
{
    if (!max_blocks_per_frame) return;
    const ZdFrameList *l = reinterpret_cast<const ZdFrameList *>(frame_lists);
    const ZdTables *t = reinterpret_cast<const ZdTables *>(tables), *p = reinterpret_cast<const ZdTables *>(predef);
    auto grid = [&](uint32_t per) { return dim3((max_blocks_per_frame + per - 1) / per, nframes); };
    if (codec == 1 && row <= 512)
        hipLaunchKernelGGL((k_block_decode<1, true, 256, 20480>), grid(256), dim3(256), 0, s, data, l, t, p, out, out_base, err, produced_out);
    else if (codec == 1)
        hipLaunchKernelGGL((k_block_decode<1, false, 64, 55296>), grid(64), dim3(64), 0, s, data, l, t, p, out, out_base, err, produced_out);
    else
        hipLaunchKernelGGL((k_block_decode<2, true, 128, 32768>), grid(128), dim3(128), 0, s, data, l, t, p, out, out_base, err, produced_out);
}
// compact lists (k_bitmap_decode_c): codec 1 zstd, 2 LZ4; nb = bytes of one binary map
void launch_bitmap_decode_compact(int codec, const uint8_t *data, const void *frame_lists, const uint64_t *src_base, uint32_t nframes,
                                  uint32_t max_blocks_per_frame, const void *tables, const void *predef, uint8_t *out, const uint64_t *out_base,
                                  uint64_t nb, int *err, hipStream_t s)
{
    if (!max_blocks_per_frame) return;
    const ZdFrameList *l = reinterpret_cast<const ZdFrameList *>(frame_lists);
    const ZdTables *t = reinterpret_cast<const ZdTables *>(tables), *p = reinterpret_cast<const ZdTables *>(predef);
    if (codec == 1)
        hipLaunchKernelGGL((k_bitmap_decode_c<1, 256, 20480>), dim3((max_blocks_per_frame + 255) / 256, nframes), dim3(256), 0, s, data, l, src_base, t, p, out,
                           out_base, nb, err);
    else
        hipLaunchKernelGGL((k_bitmap_decode_c<2, 128, 32768>), dim3((max_blocks_per_frame + 127) / 128, nframes), dim3(128), 0, s, data, l, src_base, t, p, out,
                           out_base, nb, err);
}
// lists: nlists block lists with nblocks entries in all
void launch_block_copy(const uint8_t *data, const void *lists, uint32_t nlists, uint32_t nblocks, uint32_t max_regen, uint8_t *out,
                       const uint64_t *out_base, hipStream_t s)
{
    if (!nblocks) return;
    const uint32_t ppb = std::max(1u, std::min(64u, (max_regen + 16383u) / 16384u));
    const uint32_t waves = nblocks * ppb;
    hipLaunchKernelGGL(k_block_copy, dim3((waves + WAVES - 1) / WAVES), dim3(WG), 0, s, data, reinterpret_cast<const ZdFrameList *>(lists), nlists,
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
