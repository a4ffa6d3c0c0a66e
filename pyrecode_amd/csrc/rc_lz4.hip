// rc_lz4.hip - LZ4 block encoder for packed binary maps, one wavefront per block (gfx950).
//
// Replaces the reference's `lz4.frame.compress(data, compression_level, store_size=False)` call on the packed
// binary map (pyrecode/recode_compressors.py:91, called from recode_writer.py:503-505).  The reference pins no
// compressed bytes for this scheme: the contract is a valid LZ4 frame that any stock decoder
// expands to the bit-exact bitmap.  Format: lz4_Block_format.md / lz4_Frame_format.md (v1.6.x).
//
// Encoder (data-parallel, no hash table - the input is a sparse bitmap, >90 % zero bytes at the target sparsity):
//   every run of >= 5 zero bytes becomes  [literal 0x00][match offset=1, length=run-1]  (an overlapping copy, the
//   format's RLE idiom); everything else is literals.  Sequences are found with bit-parallel mask arithmetic, their
//   encoded sizes are prefix-summed across the wavefront, and every lane writes the bytes of the positions it owns.
//   Block-end rules (last 5 bytes literal, last match starts >= 12 bytes before the end) are met by never matching
//   inside the last 12 bytes.  A block that would not shrink is stored raw (bit 31 of the block size word).
#include "rc_launch.h"

namespace rc {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// bit j (0..3) set iff byte j of x is zero (exact, no borrow artefacts)
__device__ __forceinline__ uint32_t zero_bytes4(uint32_t x)
{
    const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 in every zero byte
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}

__device__ __forceinline__ uint32_t lz4_ext(uint32_t x) { return x < 15 ? 0u : 1u + (x - 15u) / 255u; }

// smallest q >= p with bit q set in the block's match mask, or n if there is none
__device__ __forceinline__ uint32_t next_set(const uint32_t *mw, uint32_t p, uint32_t n)
{
    uint32_t wi = p >> 5;
    if (wi >= 64) return n;
    uint32_t bits = mw[wi] & (0xFFFFFFFFu << (p & 31));
    while (bits == 0) {
        if (++wi >= 64) return n;
        bits = mw[wi];
    }
    return wi * 32 + (uint32_t)__builtin_ctz(bits);
}
// smallest q >= p with bit q clear (mw[64] == 0 is a sentinel, so q <= 2048)
__device__ __forceinline__ uint32_t next_clear(const uint32_t *mw, uint32_t p)
{
    uint32_t wi = p >> 5;
    uint32_t bits = ~mw[wi] & (0xFFFFFFFFu << (p & 31));
    while (bits == 0) bits = ~mw[++wi];
    return wi * 32 + (uint32_t)__builtin_ctz(bits);
}

__device__ __forceinline__ uint32_t emit_len(uint8_t *out, uint32_t o, uint32_t r)
{
    while (r >= 255) { out[o++] = 255; r -= 255; }
    out[o++] = (uint8_t)r;
    return o;
}

// grid (ceil(ntiles/WAVES), B); wave w of the workgroup encodes block t = blockIdx.x*WAVES + w of frame blockIdx.y:
// bitmap bytes [t*TILE_BM, t*TILE_BM + n) -> blk_slots[f][t] = [u32 LZ4F block-size word][payload], blk_size = 4 + payload.
__global__ __launch_bounds__(WG) void k_lz4_bitmap(Scratch sc, uint32_t B)
{
    __shared__ uint32_t s_m[WAVES][66];
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[WAVES][TILE_BM];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[WAVES][TILE_BM + 32];

    const int w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t f = blockIdx.y;
    const uint32_t t = blockIdx.x * WAVES + w;
    if (t >= sc.ntiles) return;
    const uint64_t b0 = (uint64_t)t * TILE_BM;
    const uint32_t n = (uint32_t)min((uint64_t)TILE_BM, sc.nb - b0);
    const uint8_t *src = sc.bitmap + (uint64_t)f * sc.nb_stride + b0;  // rows are padded to whole tiles

    // this lane owns block positions [32*lane, 32*lane + 32)
    const u32x4 a = reinterpret_cast<const u32x4 *>(src)[2 * lane];
    const u32x4 b = reinterpret_cast<const u32x4 *>(src)[2 * lane + 1];
    reinterpret_cast<u32x4 *>(s_raw[w])[2 * lane] = a;
    reinterpret_cast<u32x4 *>(s_raw[w])[2 * lane + 1] = b;
    uint32_t z = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        z |= zero_bytes4(a[k]) << (4 * k);
        z |= zero_bytes4(b[k]) << (16 + 4 * k);
    }
    const int base = 32 * lane;
    auto below = [&](int lim) -> uint32_t {  // mask of own positions p < lim
        const int rel = lim - base;
        return rel >= 32 ? 0xFFFFFFFFu : (rel <= 0 ? 0u : ((1u << rel) - 1u));
    };
    const uint32_t valid = below((int)n);
    const uint32_t zeff = z & below((int)n - 12);  // never match inside the last 12 bytes

    // bit-parallel: positions inside zero runs of length >= 5, minus the first position of each run
    uint32_t prev = __shfl_up(zeff, 1), next = __shfl_down(zeff, 1);
    if (lane == 0) prev = 0;
    if (lane == 63) next = 0;
    const uint64_t W = (uint64_t)(prev >> 24) | ((uint64_t)zeff << 8) | ((uint64_t)(next & 0xFFu) << 40);
    const uint64_t R5 = W & (W >> 1) & (W >> 2) & (W >> 3) & (W >> 4);
    const uint64_t Q = R5 | (R5 << 1) | (R5 << 2) | (R5 << 3) | (R5 << 4);
    const uint64_t Mw = Q & (Q << 1);
    const uint32_t m = (uint32_t)(Mw >> 8);       // own positions that are produced by a match
    const uint32_t pm1 = (uint32_t)(Mw >> 7) & 1u;  // position base-1 is a match position
    const uint32_t after = (m << 1) | pm1;          // bit j: position j-1 is a match position
    const uint32_t lit = ~m & valid;
    const uint32_t fl = lit & (after | (lane == 0 ? 1u : 0u));  // first literal of a sequence
    const uint32_t ms = m & ~after;                             // first position of a match

    uint32_t *mw = s_m[w];
    mw[lane] = m;
    if (lane < 2) mw[64 + lane] = 0;

    // pass 1: bytes this lane will emit
    uint32_t E = (uint32_t)__builtin_popcount(lit);
    for (uint32_t bits = fl; bits; bits &= bits - 1) {
        const uint32_t p = base + (uint32_t)__builtin_ctz(bits);
        E += 1 + lz4_ext(next_set(mw, p + 1, n) - p);
    }
    for (uint32_t bits = ms; bits; bits &= bits - 1) {
        const uint32_t q = base + (uint32_t)__builtin_ctz(bits);
        E += 2 + lz4_ext(next_clear(mw, q + 1) - q - 4);
    }
    const uint32_t inc = wave_incl_scan(E);
    const uint32_t total = wave_last(inc);

    uint8_t *slot = sc.blk_slots + ((uint64_t)f * sc.ntiles + t) * BLK_SLOT;
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(slot);
    const uint32_t *payload;
    uint32_t nbytes, word;
    if (total >= n) {  // would not shrink: stored block
        payload = reinterpret_cast<const uint32_t *>(s_raw[w]);
        nbytes = n;
        word = n | 0x80000000u;
    } else {
        uint8_t *out = s_out[w];
        const uint8_t *raw = s_raw[w];
        uint32_t o = inc - E;
        for (uint32_t work = lit | ms; work; work &= work - 1) {
            const uint32_t j = (uint32_t)__builtin_ctz(work);
            const uint32_t p = base + j;
            if ((fl >> j) & 1u) {
                const uint32_t q = next_set(mw, p + 1, n);
                const uint32_t ll = q - p;
                uint32_t tok = min(ll, 15u) << 4;
                if (q < n) tok |= min(next_clear(mw, q + 1) - q - 4, 15u);
                out[o++] = (uint8_t)tok;
                if (ll >= 15) o = emit_len(out, o, ll - 15);
            }
            if ((lit >> j) & 1u) {
                out[o++] = raw[p];
            } else {  // match start: offset 1, then the length extension
                const uint32_t ml4 = next_clear(mw, p + 1) - p - 4;
                out[o++] = 1;
                out[o++] = 0;
                if (ml4 >= 15) o = emit_len(out, o, ml4 - 15);
            }
        }
        payload = reinterpret_cast<const uint32_t *>(out);
        nbytes = total;
        word = total;
    }
    if (lane == 0) {
        slot32[0] = word;
        sc.blk_size[(uint64_t)f * sc.ntiles + t] = 4 + nbytes;
    }
    for (uint32_t i = lane; i < (nbytes + 3) / 4; i += 64) slot32[1 + i] = payload[i];
}

// ---- stand-alone LZ4 frame of an arbitrary byte buffer (seam 2: compress(), recode_compressors.py:91) -----------------
// Blocks were encoded by k_lz4_bitmap with B == 1 (the buffer plays the role of one frame's bitmap); this kernel
// concatenates them behind the 7-byte frame header and appends the EndMark.  One wavefront per block.
__global__ __launch_bounds__(WG) void k_lz4f_gather(Scratch sc, uint32_t hdr3, uint8_t *__restrict__ out)
{
    const uint32_t t = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (t >= sc.ntiles) return;
    if (t == 0 && lane_id() == 0) {
        out[0] = 0x04; out[1] = 0x22; out[2] = 0x4D; out[3] = 0x18;
        out[4] = (uint8_t)hdr3; out[5] = (uint8_t)(hdr3 >> 8); out[6] = (uint8_t)(hdr3 >> 16);
        uint8_t *e = out + 7 + sc.frame_cbytes[0];
        e[0] = e[1] = e[2] = e[3] = 0;
    }
    const uint8_t *src = sc.blk_slots + (uint64_t)t * BLK_SLOT;
    uint8_t *dst = out + 7 + sc.blk_off[t];
    const uint32_t n = sc.blk_size[t];
    for (uint32_t i = lane_id(); i < n; i += 64) dst[i] = src[i];
}
void launch_lz4f_gather(const Scratch &sc, uint32_t hdr3, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_lz4f_gather, dim3((sc.ntiles + WAVES - 1) / WAVES), dim3(WG), 0, s, sc, hdr3, out);
}

// ---- LZ4 frame decode (seam 2: de_compress(), recode_compressors.py:49) -------------------------------------------------
// The host walks the frame's block headers (cheap, sequential by format) and hands over a block table; one thread
// decodes one block.  Independent blocks (what this library writes: 1024 blocks per 4096x4096 bitmap) decode in
// parallel; a frame with linked blocks (liblz4's default) is decoded by a single thread in block order.
// pass 1 (sizes != nullptr): decoded size of each block; pass 2: decode to dst + dst_off[b].
__device__ uint32_t lz4_block_walk(const uint8_t *__restrict__ src, uint32_t n, uint8_t *__restrict__ dst_base, uint64_t op0,
                                   uint64_t cap, bool write, int *err)
{
    uint32_t ip = 0;
    uint64_t op = op0;
    while (ip < n) {
        const uint32_t token = src[ip++];
        uint32_t lit = token >> 4;
        if (lit == 15) {
            uint32_t b;
            do {
                if (ip >= n) { *err = 1; return 0; }
                b = src[ip++];
                lit += b;
            } while (b == 255);
        }
        if (ip + lit > n || op + lit > cap) { *err = 1; return 0; }
        if (write)
            for (uint32_t i = 0; i < lit; ++i) dst_base[op + i] = src[ip + i];
        ip += lit;
        op += lit;
        if (ip >= n) break;
        if (ip + 2 > n) { *err = 1; return 0; }
        const uint32_t off = src[ip] | ((uint32_t)src[ip + 1] << 8);
        ip += 2;
        uint32_t ml = token & 15;
        if (ml == 15) {
            uint32_t b;
            do {
                if (ip >= n) { *err = 1; return 0; }
                b = src[ip++];
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (off == 0 || off > op || op + ml > cap) { *err = 1; return 0; }
        if (write)
            for (uint32_t i = 0; i < ml; ++i) dst_base[op + i] = dst_base[op + i - off];
        op += ml;
    }
    return (uint32_t)(op - op0);
}

__global__ void k_lz4_decode(const uint8_t *__restrict__ src, const Lz4Block *__restrict__ blks, uint32_t nblk,
                             uint32_t *__restrict__ sizes, const uint64_t *__restrict__ dst_off, uint8_t *__restrict__ dst,
                             uint64_t cap, int linked, int *__restrict__ err)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (linked) {
        if (b != 0) return;
        uint64_t op = 0;
        for (uint32_t k = 0; k < nblk; ++k) {
            const Lz4Block q = blks[k];
            if (q.raw) {
                if (op + q.size > cap) { *err = 1; return; }
                if (dst) for (uint32_t i = 0; i < q.size; ++i) dst[op + i] = src[q.src_off + i];
                op += q.size;
            } else {
                // sizes pass of a linked frame still needs the bytes (matches reach into earlier blocks) -> decode for real
                op += lz4_block_walk(src + q.src_off, q.size, dst, op, cap, dst != nullptr, err);
                if (*err) return;
            }
        }
        if (sizes) sizes[0] = (uint32_t)op;
        return;
    }
    if (b >= nblk) return;
    const Lz4Block q = blks[b];
    if (sizes) {
        int e = 0;
        // an independent block may not reference bytes before its own start: walk with op0 = 0 and an unbounded cap
        sizes[b] = q.raw ? q.size : lz4_block_walk(src + q.src_off, q.size, nullptr, 0, ~0ull, false, &e);
        if (e) *err = 1;
        return;
    }
    const uint64_t o = dst_off[b];
    if (q.raw) {
        if (o + q.size > cap) { *err = 1; return; }
        for (uint32_t i = 0; i < q.size; ++i) dst[o + i] = src[q.src_off + i];
    } else {
        int e = 0;
        (void)lz4_block_walk(src + q.src_off, q.size, dst + o, 0, cap - o, true, &e);
        if (e) *err = 1;
    }
}
void launch_lz4_decode(const uint8_t *src, const Lz4Block *blks, uint32_t nblk, uint32_t *sizes, const uint64_t *dst_off,
                       uint8_t *dst, uint64_t cap, int linked, int *err, hipStream_t s)
{
    const uint32_t threads = 64, grid = linked ? 1 : (nblk + threads - 1) / threads;
    hipLaunchKernelGGL(k_lz4_decode, dim3(grid), dim3(threads), 0, s, src, blks, nblk, sizes, dst_off, dst, cap, linked, err);
}

void launch_lz4_encode_bitmap(const Scratch &sc, uint32_t B, hipStream_t s)
{
    const dim3 grid((sc.ntiles + WAVES - 1) / WAVES, B), block(WG);
    hipLaunchKernelGGL(k_lz4_bitmap, grid, block, 0, s, sc, B);
}

}  // namespace rc
