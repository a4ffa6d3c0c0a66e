// rc_pix_huff.hip - entropy coding of the residual-intensity stream for the modelled zstd encoder (gfx950).
//
// Replaces the second `compress()` of the reference's per-frame step, the one on the packed pixel intensities
// (pyrecode/recode_writer.py:507-511 -> recode_compressors.py:88, ZstdCompressor.compress).  The packed stream has no
// repeats to speak of (stock libzstd finds none either: its 0.80 on the bench data is all Huffman), so the encoder is the
// literal half of a zstd block only: the frame's stream, laid out flat by k_gather (pix_mode 1), is cut into chunks of
// PIX_CHUNK bytes; every chunk becomes one block of Huffman-coded literals without sequences (single stream, treeless: the
// tree of the ctx's model travels in the frame's first such block, k_pix_scan puts it there), or a Raw block when that would
// not be smaller.  Serial specification of a chunk: zm_encode_pix_chunk (rc_zstd_block.h), judged by stock libzstd in
// tests/test_zstd_format_cpu.py.
//
//   k_pix_huff    one wavefront per chunk (a fixed number of wavefronts per frame loop over its chunks): 16 bytes per lane,
//                 code lookups in an LDS copy of the table, one DPP scan for the bit offsets (the LAST byte sits in the
//                 lowest bits), LDS atomic ORs build the stream, coalesced dword stores write the chunk's slot
//   k_pix_scan    (rc_reduce.hip, next to k_scan_frames whose helpers it shares) sizes -> offsets, tree into the first chunk
//   k_pix_gather  copies the chunks behind the frame header of the record's residual stream
#include "rc_launch.h"
#include "rc_lz4_block.h"
#include "rc_zstd_block.h"
#include "rc_zstd_wave.h"

namespace rc {

constexpr int PH_WG_PER_FRAME = 16;   // x WAVES wavefronts loop over a frame's chunks (333 chunks per 4096^2 frame at 1 %)
constexpr int PH_DW = 360;            // LDS dwords per wavefront: 6 + 1008 * 11 / 8 = 1392 bytes + slack for the OR window

__device__ __forceinline__ uint32_t pix_packed_bytes(uint32_t nnz, uint32_t depth)
{
    return depth == 16 ? nnz * 2u : (uint32_t)(((uint64_t)nnz * depth + 7) >> 3);
}

__global__ __launch_bounds__(WG) void k_pix_huff(Scratch sc, uint32_t B, uint32_t depth)
{
    __shared__ uint16_t s_code[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_out[WAVES][PH_DW];
    const ZstdModel *M = reinterpret_cast<const ZstdModel *>(sc.zm_model);
    if (threadIdx.x < 128) reinterpret_cast<uint32_t *>(s_code)[threadIdx.x] = reinterpret_cast<const uint32_t *>(M->pix_code)[threadIdx.x];
    __syncthreads();
    const uint32_t f = blockIdx.y;
    if (f >= B) return;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t npk = pix_packed_bytes(sc.frame_nnz[f], depth);
    const uint32_t nch = npk ? (npk + PIX_CHUNK - 1) / PIX_CHUNK : 1u;
    const bool can = (sc.zm_valid & 2u) != 0;
    const uint32_t desc_len = M->pix_desc_len;
    uint32_t *out32 = s_out[w];
    uint8_t *out8 = reinterpret_cast<uint8_t *>(out32);
    for (uint32_t c = blockIdx.x * WAVES + w; c < nch; c += gridDim.x * WAVES) {
        const uint32_t n = min(PIX_CHUNK, npk - c * PIX_CHUNK);   // (npk == 0: one empty Raw block)
        const uint32_t lastbit = c + 1 == nch ? 1u : 0u;
        const uint8_t *src = sc.pixraw + (uint64_t)f * sc.pixraw_stride + (uint64_t)c * PIX_CHUNK;
        const int vb = max(0, min(16, (int)n - 16 * lane));     // this lane's valid bytes
        u32x4 v = {0u, 0u, 0u, 0u};
        if (vb > 0) v = *reinterpret_cast<const u32x4 *>(src + 16 * lane);   // rows and chunks are 16-byte aligned and padded
        // codes of the lane's bytes, in groups of four (<= 44 bits): inside a group the LAST byte is lowest
        uint64_t g[4];
        uint32_t gb[4], nb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint64_t a = 0;
            uint32_t b = 0;
#pragma unroll
            for (int j = 3; j >= 0; --j)
                if (4 * k + j < vb) {
                    const uint32_t cd = s_code[(v[k] >> (8 * j)) & 0xFFu];
                    a |= (uint64_t)(cd & 0xFFFu) << b;
                    b += cd >> 12;
                }
            g[k] = a; gb[k] = b; nb += b;
        }
        const uint32_t binc = wave_incl_scan(nb);
        const uint32_t hbits = wave_last(binc);
        const uint32_t hbytes = (hbits + 8) >> 3;
        const uint32_t content = 3 + hbytes + 1;
        const bool comp = can && n > desc_len + 8u && content < n - desc_len - 8u;   // (zm_encode_pix_chunk's rule)
        uint32_t size;
        if (comp) {
            const uint32_t ndw = (6 + hbytes + 1 + 3) >> 2;
            for (uint32_t i = lane; i < ndw + 3; i += 64) out32[i] = 0;
            __builtin_amdgcn_wave_barrier();
            uint32_t bit = 48 + (hbits - binc);   // the lane's LAST group first
#pragma unroll
            for (int k = 3; k >= 0; --k)
                if (gb[k]) {
                    const uint32_t wd = bit >> 5, s = bit & 31u;
                    const uint64_t a = g[k] << s;
                    const uint32_t top = s ? (uint32_t)(g[k] >> (64 - s)) : 0u;
                    __hip_atomic_fetch_or(&out32[wd], (uint32_t)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (a >> 32) __hip_atomic_fetch_or(&out32[wd + 1], (uint32_t)(a >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (top) __hip_atomic_fetch_or(&out32[wd + 2], top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    bit += gb[k];
                }
            if (lane == 0) {
                const uint32_t em = 48 + hbits;
                __hip_atomic_fetch_or(&out32[em >> 5], 1u << (em & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const uint32_t bh = lastbit | (2u << 1) | (content << 3);          // Block_Header
                const uint32_t lh = 3u | (n << 4) | (hbytes << 14);                 // treeless, single stream, 10-bit sizes
                __hip_atomic_fetch_or(&out32[0], bh | (lh << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_or(&out32[1], lh >> 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            size = 3 + content;   // (the Number_of_Sequences byte behind the stream is one of the zeroed bytes)
        } else {   // Raw block
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const uint32_t bh = lastbit | (n << 3);
                out8[0] = (uint8_t)bh; out8[1] = (uint8_t)(bh >> 8); out8[2] = (uint8_t)(bh >> 16);
            }
            for (int i = 0; i < vb; ++i) out8[3 + 16 * lane + i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
            size = 3 + n;
        }
        __builtin_amdgcn_wave_barrier();
        const uint64_t fc = (uint64_t)f * sc.nchunk_max + c;
        uint32_t *slot = reinterpret_cast<uint32_t *>(sc.pix_chunks + fc * PIX_SLOT);
        for (uint32_t i = lane; i < (size + 3) >> 2; i += 64) slot[i] = out32[i];
        if (lane == 0) sc.chunk_size[fc] = size | (comp ? ZW_TREE : 0u);
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_pix_huff(const Scratch &sc, uint32_t B, uint32_t depth, hipStream_t s)
{
    hipLaunchKernelGGL(k_pix_huff, dim3(PH_WG_PER_FRAME, B), dim3(WG), 0, s, sc, B, depth);
}

// one wavefront per chunk copy; chunk 0's wavefront also writes the stream's frame header and the two record fields
__global__ __launch_bounds__(WG) void k_pix_gather(Scratch sc, uint32_t B, uint32_t depth, uint32_t rec_hdr, uint8_t *__restrict__ out,
                                                     const uint64_t *__restrict__ rec_off)
{
    const uint32_t f = blockIdx.y;
    if (f >= B || sc.status->code != 0) return;
    const int lane = lane_id();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t npk = pix_packed_bytes(sc.frame_nnz[f], depth);
    const uint32_t nch = npk ? (npk + PIX_CHUNK - 1) / PIX_CHUNK : 1u;
    // the residual stream starts behind the record header and the bitmap stream: [6-byte frame header][blocks]
    uint8_t *rec = out + rec_off[f];
    const uint32_t cb = 6 + sc.frame_cbytes[f];   // the bitmap stream: frame header + its blocks
    uint8_t *pf = rec + rec_hdr + cb;
    for (uint32_t c = blockIdx.x * WAVES + w; c < nch; c += gridDim.x * WAVES) {
        const uint64_t fc = (uint64_t)f * sc.nchunk_max + c;
        const uint32_t size = sc.chunk_size[fc], off = sc.chunk_off[fc];
        const uint8_t *src = sc.pix_chunks + fc * PIX_SLOT;   // 4-byte aligned (slots are PIX_SLOT apart)
        uint8_t *dst = pf + 6 + off;                          // any alignment
        // aligned dword stores: destination dword j = source bytes [head + 4j, +4) = the byte funnel of source dwords j, j+1
        const uint32_t head = min(size, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));
        const uint32_t nd = (size - head) >> 2, tail = (size - head) & 3u;
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
        for (uint32_t j = lane; j < nd; j += 64)
            reinterpret_cast<uint32_t *>(dst + head)[j] = __builtin_amdgcn_alignbyte(s32[j + 1], s32[j], head);
        if ((uint32_t)lane < head) dst[lane] = src[lane];
        if ((uint32_t)lane < tail) dst[head + 4 * nd + lane] = src[head + 4 * nd + lane];
        if (c == 0 && lane == 0) {   // magic, Frame_Header_Descriptor 0, 1 KiB window (blocks regenerate <= 1008 bytes)
            pf[0] = 0x28; pf[1] = 0xB5; pf[2] = 0x2F; pf[3] = 0xFD; pf[4] = 0; pf[5] = 0;
        }
    }
}

void launch_pix_gather(const Scratch &sc, uint32_t B, uint32_t depth, uint32_t rec_hdr, uint8_t *out, const uint64_t *rec_off,
                       hipStream_t s)
{
    hipLaunchKernelGGL(k_pix_gather, dim3(PH_WG_PER_FRAME, B), dim3(WG), 0, s, sc, B, depth, rec_hdr, out, rec_off);
}

}  // namespace rc
