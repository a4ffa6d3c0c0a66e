// rc_reduce32.hip - the reduce step (A1-A5) for sources beyond 16 bits: uint32 frames and dark frame (gfx950).
//
// The reference's Python path takes whatever map_dtype yields for (source_data_type, source_bit_depth): uint32 above 16 bits
// (pyrecode/misc.py:41-49); thr, `frame > thr` and the residuals are then uint32 arithmetic (recode_writer.py:126-137,437-440) and the
// residual stream is `source_bit_depth`-bit fields, LSB first (_bit_pack, :637-652) - or the values' four raw bytes when the depth is a
// multiple of 8 (`.tobytes()`, :463-464: 32 AND 24; the caller passes 32 for both).
//
// Not the hot path of any BASELINE configuration (all of them are uint16): one straightforward kernel, a wavefront per tile and frame,
// no register-resident threshold, no rolling loads, no fused codec.  It leaves what rc_reduce.hip's kernel leaves - the raw binary map
// (always: the block encoders run over it as separate launches, rc_lz4.hip / rc_zstd.hip / rc_blosc.hip), per tile the packed residual
// stream in its slot (whole 128-byte lines, zero behind the last field) and the count - so scans, record layout and assembly are the
// uint16 path's, unchanged (k_assemble concatenates bit streams of any field width up to 32).
#include "rc_launch.h"

namespace rc {

constexpr int R32_WAVES = 1;   // wavefronts per workgroup: 32.5 KB of LDS each (four of them share a CU)
struct __attribute__((aligned(16))) Stage32 {
    uint32_t val[TILE_PX];     // the tile's residuals in pixel order
    uint32_t out[TILE_PX];     // compacted, then packed in place
    uint8_t bm[TILE_BM];       // bitmap bytes, transposed so that a lane owns 64 consecutive pixels
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

// grid (ceil(ntiles / R32_WAVES), B)
__global__ __launch_bounds__(64 * R32_WAVES) void k_reduce_tiles32(const uint32_t *__restrict__ frames, const uint32_t *__restrict__ thr, uint64_t N,
                                                                     uint32_t ntiles, uint8_t *__restrict__ bitmap, uint64_t nb_stride,
                                                                     uint8_t *__restrict__ pix_slots, uint32_t pix_slot_bytes,
                                                                     uint32_t *__restrict__ tile_cnt, uint32_t depth, uint32_t level1,
                                                                     BatchStatus *__restrict__ status)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { status->code = 0; status->frame = 0; status->total = 0; }
    __shared__ Stage32 s_st[R32_WAVES];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t tile = blockIdx.x * R32_WAVES + w, f = blockIdx.y;
    if (tile >= ntiles) return;
    Stage32 &S = s_st[w];
    const uint32_t *fr = frames + (uint64_t)f * N;
    // A2 + A3: eight pixels per lane and group (one bitmap byte), residuals staged in pixel order
#pragma unroll 2
    for (int r = 0; r < R; ++r) {
        const uint64_t px0 = (uint64_t)tile * TILE_PX + (uint64_t)r * GROUP_PX + (uint64_t)lane * 8;
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t v = 0;
            if (px0 + k < N) {
                const uint32_t a = fr[px0 + k], t = thr[px0 + k];
                if (a > t) { v = a - t; m |= 1u << k; }
            }
            S.val[r * GROUP_PX + lane * 8 + k] = v;
        }
        S.bm[r * 64 + lane] = (uint8_t)m;
    }
    __builtin_amdgcn_wave_barrier();
    // A4: after the transpose a lane owns 8 consecutive bitmap bytes = 64 consecutive pixels
    const u32x2 own = *reinterpret_cast<const u32x2 *>(&S.bm[lane * 8]);
    *reinterpret_cast<u32x2 *>(bitmap + (uint64_t)f * nb_stride + (uint64_t)tile * TILE_BM + lane * 8) = own;
    if (!level1) return;
    const uint64_t ft = (uint64_t)f * ntiles + tile;
    const uint32_t cnt = (uint32_t)__builtin_popcount(own[0]) + (uint32_t)__builtin_popcount(own[1]);
    const uint32_t inc = wave_incl_scan(cnt);
    const uint32_t total = wave_last(inc);
    {   // row-major order is lane order: every lane moves its own set pixels
        uint64_t q = (uint64_t)own[0] | ((uint64_t)own[1] << 32);
        uint32_t e = inc - cnt;
        const uint32_t *mine = S.val + 64 * lane;
        for (; q; q &= q - 1) S.out[e++] = mine[__builtin_ctzll(q)];
    }
    __builtin_amdgcn_wave_barrier();
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
    uint32_t *dst = reinterpret_cast<uint32_t *>(pix_slots + ft * pix_slot_bytes);
    for (uint32_t j = lane; j < ndw; j += 64) dst[j] = S.out[j];
    if (lane == 0) tile_cnt[ft] = total;
}

void launch_reduce32(const Scratch &sc, const uint32_t *frames, const uint32_t *thr32, uint32_t B, uint32_t level, uint32_t depth, hipStream_t s)
{
    const dim3 grid((sc.ntiles + R32_WAVES - 1) / R32_WAVES, B);
    hipLaunchKernelGGL(k_reduce_tiles32, grid, dim3(64 * R32_WAVES), 0, s, frames, thr32, sc.N, sc.ntiles, sc.bitmap, sc.nb_stride,
                       reinterpret_cast<uint8_t *>(sc.pix_slots), sc.pix_slot_bytes, sc.tile_cnt, depth, level == 1 ? 1u : 0u, sc.status);
}

}  // namespace rc
