// rc_reduce32.hip - the reduce step (A1-A5) for sources beyond 16 bits: uint32 frames and dark frame (gfx950).
//
// The reference's Python path takes whatever map_dtype yields for (source_data_type, source_bit_depth): uint32 above 16 bits
// (pyrecode/misc.py:41-49); thr, `frame > thr` and the residuals are then uint32 arithmetic (recode_writer.py:126-137,437-440) and the
// residual stream is `source_bit_depth`-bit fields, LSB first (_bit_pack, :637-652) - or the values' four raw bytes when the depth is a
// multiple of 8 (`.tobytes()`, :463-464: 32 AND 24; the caller passes 32 for both).
//
// Not the hot path of any BASELINE configuration (all of them are uint16), but built the way the uint16 kernel is: a wavefront keeps its
// tile's thresholds in registers (64 VGPRs) over FPW consecutive frames, the next frame's 16 loads of 16 bytes per lane are in flight
// while the current one is reduced (two register sets), loads are whole 1 KiB runs per instruction (lane l: pixels 4 l .. 4 l + 3 of the
// group's first and second half), compaction goes from the registers into LDS (one packed wave scan per group), the d-bit pack in place.
// The block encoders - LZ4, blosc-lz4 (rc_lz4_block.h), the zstd tokenizer (rc_zstd_wave.h, the fast form; k_zstd_fse finishes its blocks) -
// run inside the kernel on the tile's map in LDS, as in the uint16 kernel; reduce-only records take the raw binary map.  Either
// way it leaves what rc_reduce.hip's kernel leaves - per tile the encoded block (or the map), the packed residual stream in its slot (whole
// 128-byte lines, zero behind the last field) and the count - so scans, record layout and assembly are the uint16 path's, unchanged
// (k_gather concatenates bit streams of any field width up to 32).  Algorithmic bytes: 4 N per frame in; blocks and residual lines out.
#include <type_traits>

#include "rc_launch.h"
#include "rc_lz4_block.h"
#include "rc_zstd_wave.h"

namespace rc {

#ifndef RC_R32_WAVES
#define RC_R32_WAVES 1
#endif
constexpr int R32_WAVES = RC_R32_WAVES;   // wavefronts per workgroup (each with its own tile and LDS stage); 1 stays: 2 / 4 per workgroup ran LZ4 6-8 % slower
                                          // (72.6 k -> 68.0 / 67.1 k frames/s), zstd the same (same box, -DRC_R32_WAVES)
constexpr int R32_FPW = 16;    // frames a wavefront keeps its tile for (the thresholds are read once per R32_FPW frames)
struct __attribute__((aligned(16))) Stage32 {
    uint32_t out[TILE_PX];     // compacted residuals in pixel order, then packed in place
    Lz4Lds lz;                 // lz.raw: the tile's bitmap bytes in pixel order = the block image of the LZ4 encoder (fused: CODEC 2 / 4)
};

__global__ void k_threshold32(const uint32_t *__restrict__ dark, uint32_t eps, uint64_t N, uint32_t *__restrict__ thr)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < N; i += stride) thr[i] = dark[i] + eps;   // wraps mod 2^32 like NumPy 2's uint32 + python int
}
void launch_threshold32(const uint32_t *dark, int64_t eps, uint64_t N, uint32_t *thr, hipStream_t s)
{
    uint32_t blocks = (uint32_t)((N + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_threshold32, dim3(blocks), dim3(256), 0, s, dark, (uint32_t)(uint64_t)eps, N, thr);
}

// One tile's pixels of one frame (or of the threshold frame) in registers: group g's first half in v[2 g] (lane l: pixels 4 l .. 4 l + 3
// of the half), its second half in v[2 g + 1].  FULL: the tile lies wholly inside the frame - 16-byte loads (dword alignment is all a
// global load needs); the partial last tile: guarded single loads, `fill` elsewhere.
template <bool FULL, bool NT>
__device__ __forceinline__ void load_tile32(const uint32_t *__restrict__ base, uint64_t px0, uint64_t N, uint32_t fill, u32x4 (&v)[2 * R])
{
    const int lane = lane_id();
#pragma unroll
    for (int i = 0; i < 2 * R; ++i) {
        if (FULL) {
            typedef u32x4 u32x4_dw __attribute__((aligned(4)));   // (frames start on dwords, not on 16-byte boundaries)
            const u32x4_dw *p = reinterpret_cast<const u32x4_dw *>(base + px0 + (uint64_t)i * 256 + (uint64_t)lane * 4);
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        } else {
            const uint64_t px = px0 + (uint64_t)i * 256 + (uint64_t)lane * 4;
            if (px + 4 <= N) {   // (whole groups of four as vectors here too: single loads only where the frame ends)
                typedef u32x4 u32x4_dw __attribute__((aligned(4)));
                v[i] = *reinterpret_cast<const u32x4_dw *>(base + px);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] = px + k < N ? base[px + k] : fill;
            }
        }
    }
}

// A2-A5 of one tile of one frame from registers
// CODEC: 0 = none (only the raw map leaves), 1 = the zstd tokenizer (fast form), 2 / 4 = the LZ4 block encoder (runs / events), 8 = blosc1's
// bit-shuffle + LZ4 runs - on the tile's map
// here: the encoded block goes to the tile's block slot, the raw map only where the caller keeps binary maps (bm_dst != nullptr)
template <int CODEC>
__device__ __forceinline__ void reduce_tile32(Stage32 &S, const u32x4 (&x)[2 * R], const u32x4 (&t)[2 * R], uint8_t *__restrict__ bm_dst,
                                              uint8_t *__restrict__ slot, uint32_t *__restrict__ cnt_dst, uint32_t depth, bool level1,
                                              uint8_t *__restrict__ blk_slot, uint32_t *__restrict__ blk_size_dst, uint32_t n_blk, bool last_blk)
{
    const int lane = lane_id();
    uint32_t base = 0;
#pragma unroll
    for (int g = 0; g < R; ++g) {
        uint32_t m = 0;   // bits 0..3: first half, 4..7: second half
        u32x4 ra, rb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = x[2 * g][k], ta = t[2 * g][k], b = x[2 * g + 1][k], tb = t[2 * g + 1][k];
            m |= (a > ta ? 1u : 0u) << k | (b > tb ? 1u : 0u) << (4 + k);
            ra[k] = a - ta;
            rb[k] = b - tb;
        }
        // A4: a bitmap byte is two neighbouring lanes' nibbles - the even lane forms the first half's byte, the odd lane the second half's
        const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        const uint32_t byte = (lane & 1) ? ((other >> 4) | (m & 0xF0u)) : ((m & 0xFu) | ((other & 0xFu) << 4));
        S.lz.raw[g * 64 + (lane & 1) * 32 + (lane >> 1)] = (uint8_t)byte;
        if (level1) {
            // A3: row-major order = the first half's lanes in order, then the second half's
            const uint32_t ca = (uint32_t)__builtin_popcount(m & 0xFu), cb = (uint32_t)__builtin_popcount(m >> 4);
            const uint32_t inc = wave_incl_scan(ca | cb << 16);
            const uint32_t tot = wave_last(inc);
            if (tot) {
                uint32_t pa = base + (inc & 0xFFFFu) - ca, pb = base + (tot & 0xFFFFu) + (inc >> 16) - cb;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (m >> k & 1u) S.out[pa++] = ra[k];
                    if (m >> (4 + k) & 1u) S.out[pb++] = rb[k];
                }
            }
            base += (tot & 0xFFFFu) + (tot >> 16);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const u32x2 ownv = *reinterpret_cast<const u32x2 *>(&S.lz.raw[lane * 8]);
    if (bm_dst) *reinterpret_cast<u32x2 *>(bm_dst + lane * 8) = ownv;
    if (CODEC == 1) {   // zstd, the fast encoder's wave-collective half (k_zstd_fse finishes the block)
        uint32_t staged;
        const uint32_t word = zstd_tokenize_block((uint64_t)ownv[0] | ((uint64_t)ownv[1] << 32), n_blk, last_blk, S.lz, staged);
        zstd_store_block(blk_slot, n_blk, last_blk, word, staged, S.lz);
        if (lane == 0) *blk_size_dst = word;
        __builtin_amdgcn_wave_barrier();
    } else if (CODEC) {
        uint64_t own = (uint64_t)ownv[0] | ((uint64_t)ownv[1] << 32);
        if (CODEC == 8) own = bitshuffle_block(own, n_blk, S.lz);   // blosc1: the block's bit-shuffle in front of the LZ4 run encoder
        const uint32_t csize = lz4_encode_block<CODEC == 4>(own, n_blk, S.lz);
        const uint32_t used = lz4_store_block(blk_slot, own, n_blk, csize, S.lz, CODEC == 8);
        if (lane == 0) *blk_size_dst = used;
        __builtin_amdgcn_wave_barrier();   // S.lz is the next frame's
    }
    if (!level1) return;
    const uint32_t total = base;
    // A5: depth-bit fields, LSB first, in place (output dword j needs values from index 32 j / depth >= j on: at or behind dword j, and
    // every lane of a step reads before any of them writes); whole 128-byte lines, zero behind the last field
    const uint32_t nbits = total * depth;   // <= 4096 * 32
    const uint32_t ndw = (((nbits + 31) >> 5) + 31u) & ~31u;
    if (depth < 32) {
        const uint32_t dmask = (1u << depth) - 1u;
        for (uint32_t j0 = 0; j0 < ndw; j0 += 64) {
            const uint32_t j = j0 + lane;
            uint32_t v = (32u * j) / depth;
            const uint32_t o = 32u * j - v * depth;
            uint64_t acc = 0;
            uint32_t filled = 0;
            if (v < total) { acc = (uint64_t)(S.out[v] & dmask) >> o; filled = depth - o; ++v; }
            while (filled < 32 && v < total) {
                acc |= (uint64_t)(S.out[v] & dmask) << filled;
                filled += depth;
                ++v;
            }
            __builtin_amdgcn_wave_barrier();
            if (j < ndw) S.out[j] = (uint32_t)acc;
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (uint32_t j = total + lane; j < ndw; j += 64) S.out[j] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t *dst = reinterpret_cast<uint32_t *>(slot);
    for (uint32_t j = lane; j < ndw; j += 64) dst[j] = S.out[j];
    if (lane == 0) *cnt_dst = total;
    __builtin_amdgcn_wave_barrier();   // S.out is the next frame's
}

// FULL: grid (tiles wholly inside the frame, ceil(B / R32_FPW)), two frame register sets; the other instantiation: the partial last tile
// (tile0 = its index), one set.  One wavefront per workgroup (18.4 KB of LDS: eight of them share a CU, what the registers allow)
template <bool FULL, int CODEC>
__global__ __launch_bounds__(64 * R32_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_reduce_tiles32(const uint32_t *__restrict__ frames, const uint32_t *__restrict__ thr, uint64_t N,
                                                       uint32_t ntiles, uint32_t tile0, uint32_t tile_n, uint32_t B, uint8_t *__restrict__ bitmap, uint64_t nb_stride,
                                                       uint8_t *__restrict__ pix_slots, uint32_t pix_slot_bytes,
                                                       uint32_t *__restrict__ tile_cnt, uint32_t depth, uint32_t level1,
                                                       BatchStatus *__restrict__ status, uint8_t *__restrict__ blk_slots, uint32_t blk_stride,
                                                       uint32_t *__restrict__ blk_size, uint64_t nb)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { status->code = 0; status->frame = 0; status->total = 0; }
    __shared__ Stage32 s_st[R32_WAVES];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Stage32 &S = s_st[wv];
    const uint32_t tile = tile0 + blockIdx.x * R32_WAVES + wv;
    if (tile >= (FULL ? tile0 + tile_n : ntiles)) return;   // (no barrier anywhere: a wavefront may leave alone)
    const uint32_t f0 = blockIdx.y * R32_FPW, f1 = f0 + R32_FPW < B ? f0 + R32_FPW : B;
    const uint64_t px0 = (uint64_t)tile * TILE_PX;
    const uint32_t n_blk = (uint32_t)min((uint64_t)TILE_BM, nb - (uint64_t)tile * TILE_BM);   // bitmap bytes of this tile
    u32x4 t[2 * R], xa[2 * R];
    load_tile32<FULL, false>(thr, px0, N, 0xFFFFFFFFu, t);
    auto one = [&](const u32x4 (&x)[2 * R], uint32_t f) {
        const uint64_t ft = (uint64_t)f * ntiles + tile;
        reduce_tile32<CODEC>(S, x, t, bitmap ? bitmap + (uint64_t)f * nb_stride + (uint64_t)tile * TILE_BM : nullptr, pix_slots + ft * pix_slot_bytes,
                             tile_cnt + ft, depth, level1 != 0, blk_slots + ft * blk_stride, blk_size + ft, n_blk, tile + 1 == ntiles);
    };
    if (FULL) {
        u32x4 xb[2 * R];
        load_tile32<true, true>(frames + (uint64_t)f0 * N, px0, N, 0, xa);
        for (uint32_t f = f0; f < f1; f += 2) {
            if (f + 1 < f1) load_tile32<true, true>(frames + (uint64_t)(f + 1) * N, px0, N, 0, xb);
            one(xa, f);
            if (f + 1 < f1) {
                if (f + 2 < f1) load_tile32<true, true>(frames + (uint64_t)(f + 2) * N, px0, N, 0, xa);
                one(xb, f + 1);
            }
        }
    } else {
        for (uint32_t f = f0; f < f1; ++f) {
            load_tile32<false, true>(frames + (uint64_t)f * N, px0, N, 0, xa);
            one(xa, f);
        }
    }
}

// codec: 0 = the raw maps only (reduce-only records), 1 = zstd (fast form; k_zstd_fse follows), 2 / 4 = LZ4 runs / events, 8 = blosc-lz4
// fused; keep_bitmap: the raw maps leave as well (validation frames, rc_get_binary_map)
void launch_reduce32(const Scratch &sc, const uint32_t *frames, const uint32_t *thr32, uint32_t B, uint32_t level, uint32_t depth, hipStream_t s,
                     uint32_t codec, bool keep_bitmap)
{
    const uint32_t nfull = (uint32_t)(sc.N / TILE_PX), fy = (B + R32_FPW - 1) / R32_FPW;
    uint8_t *slots = reinterpret_cast<uint8_t *>(sc.pix_slots);
    uint8_t *bitmap = (codec == 0 || keep_bitmap) ? sc.bitmap : nullptr;
    auto go = [&](auto full, auto cd) {
        constexpr bool FULL = decltype(full)::value;
        constexpr int CODEC = decltype(cd)::value;
        hipLaunchKernelGGL((k_reduce_tiles32<FULL, CODEC>), FULL ? dim3((nfull + R32_WAVES - 1) / R32_WAVES, fy) : dim3(1, fy), dim3(64 * R32_WAVES), 0, s, frames, thr32, sc.N, sc.ntiles,
                           FULL ? 0u : nfull, FULL ? nfull : 1u, B,
                           bitmap, sc.nb_stride, slots, sc.pix_slot_bytes, sc.tile_cnt, depth, level == 1 ? 1u : 0u, sc.status, sc.blk_slots, sc.blk_stride,
                           sc.blk_size, sc.nb);
    };
    auto both = [&](auto cd) {
        if (nfull) go(std::true_type{}, cd);
        if (nfull < sc.ntiles) go(std::false_type{}, cd);
    };
    if (codec == 2) both(std::integral_constant<int, 2>{});
    else if (codec == 4) both(std::integral_constant<int, 4>{});
    else if (codec == 8) both(std::integral_constant<int, 8>{});
    else if (codec == 1) both(std::integral_constant<int, 1>{});
    else both(std::integral_constant<int, 0>{});
}

}  // namespace rc
