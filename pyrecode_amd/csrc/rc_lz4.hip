// rc_lz4.hip - stand-alone LZ4 frame encode / decode of arbitrary buffers (seam 2); block encoder in rc_lz4_block.h.
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
#include <algorithm>

#include "rc_launch.h"
#include "rc_lz4_block.h"

namespace rc {

// grid ceil(ntiles/WAVES); wave w encodes block t = blockIdx.x*WAVES + w of the buffer sc.bitmap[0..sc.nb) (padded to whole
// blocks): blk_slots[t] = [u32 LZ4F block-size word][payload], blk_size[t] = bytes used.  Same encoder as the fused
// reduce kernel (rc_lz4_block.h).
template <bool EVENTS>
__global__ __launch_bounds__(WG) void k_lz4_buffer(Scratch sc)
{
    __shared__ Lz4Lds s_lz[WAVES];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t t = blockIdx.x * WAVES + w, f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint64_t b0 = (uint64_t)t * TILE_BM;
    const uint32_t n = (uint32_t)min((uint64_t)TILE_BM, sc.nb - b0);
    const u32x2 v = reinterpret_cast<const u32x2 *>(sc.bitmap + (uint64_t)f * sc.nb_stride + b0)[lane];  // rows are padded to whole blocks
    reinterpret_cast<u32x2 *>(s_lz[w].raw)[lane] = v;
    const uint64_t own = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    const uint32_t csize = lz4_encode_block<EVENTS>(own, n, s_lz[w]);
    const uint64_t ft = (uint64_t)f * sc.ntiles + t;
    const uint32_t used = lz4_store_block(sc.blk_slots + ft * sc.blk_stride, own, n, csize, s_lz[w]);
    if (lane == 0) sc.blk_size[ft] = used;
}
void launch_lz4_encode_buffer(const Scratch &sc, hipStream_t s, bool events)
{
    if (events) hipLaunchKernelGGL(k_lz4_buffer<true>, dim3((sc.ntiles + WAVES - 1) / WAVES), dim3(WG), 0, s, sc);
    else hipLaunchKernelGGL(k_lz4_buffer<false>, dim3((sc.ntiles + WAVES - 1) / WAVES), dim3(WG), 0, s, sc);
}

void launch_lz4_encode_rows(const Scratch &sc, uint32_t B, hipStream_t s, bool events)
{
    const dim3 grid((sc.ntiles + WAVES - 1) / WAVES, B);
    if (events) hipLaunchKernelGGL(k_lz4_buffer<true>, grid, dim3(WG), 0, s, sc);
    else hipLaunchKernelGGL(k_lz4_buffer<false>, grid, dim3(WG), 0, s, sc);
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
        if (o + q.size > cap) *err = 1;   // the bytes themselves: k_lz4_copy_stored (a stored block can be megabytes; one lane is no way to move it)
    } else {
        int e = 0;
        (void)lz4_block_walk(src + q.src_off, q.size, dst + o, 0, cap - o, true, &e);
        if (e) *err = 1;
    }
}
// stored blocks of a frame of independent blocks: one wavefront per 16 KiB piece (grid.y pieces per block), 16 bytes per lane where
// source and destination allow it
__global__ __launch_bounds__(WG) void k_lz4_copy_stored(const uint8_t *__restrict__ src, const Lz4Block *__restrict__ blks, uint32_t nblk,
                                                          const uint64_t *__restrict__ dst_off, uint8_t *__restrict__ dst, uint64_t cap)
{
    const uint32_t b = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const Lz4Block q = blks[b];
    if (!q.raw) return;
    const uint64_t o = dst_off[b];
    if (o + q.size > cap) return;   // (k_lz4_decode has flagged it)
    const int lane = lane_id();
    const uint8_t *sp = src + q.src_off;
    uint8_t *dp = dst + o;
    for (uint64_t p0 = (uint64_t)blockIdx.y * 16384u; p0 < q.size; p0 += (uint64_t)gridDim.y * 16384u) {
        const uint32_t n = (uint32_t)min<uint64_t>(16384u, q.size - p0);
        if ((((uintptr_t)(sp + p0) | (uintptr_t)(dp + p0)) & 15u) == 0) {
            const u32x4 *s4 = reinterpret_cast<const u32x4 *>(sp + p0);
            u32x4 *d4 = reinterpret_cast<u32x4 *>(dp + p0);
            for (uint32_t i = lane; i < n / 16; i += 64) d4[i] = s4[i];
            for (uint32_t i = (n & ~15u) + lane; i < n; i += 64) dp[p0 + i] = sp[p0 + i];
        } else
            for (uint32_t i = lane; i < n; i += 64) dp[p0 + i] = sp[p0 + i];
    }
}

// max_stored: size of the largest stored block (0: none) - only the decoding pass (dst != NULL) of an independent-block frame uses it
void launch_lz4_decode(const uint8_t *src, const Lz4Block *blks, uint32_t nblk, uint32_t *sizes, const uint64_t *dst_off,
                       uint8_t *dst, uint64_t cap, int linked, int *err, hipStream_t s, uint32_t max_stored)
{
    const uint32_t threads = 64, grid = linked ? 1 : (nblk + threads - 1) / threads;
    hipLaunchKernelGGL(k_lz4_decode, dim3(grid), dim3(threads), 0, s, src, blks, nblk, sizes, dst_off, dst, cap, linked, err);
    if (!linked && dst && max_stored) {
        const uint32_t pieces = std::min(256u, (max_stored + 16383u) / 16384u);
        hipLaunchKernelGGL(k_lz4_copy_stored, dim3((nblk + WAVES - 1) / WAVES, pieces), dim3(WG), 0, s, src, blks, nblk, dst_off, dst, cap);
    }
}

}  // namespace rc
