// rc_device.h - shared device-side helpers and geometry for librecode_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Experiment knobs read from the environment exist only in development builds (-DRC_DEV_KNOBS: tools/build_def.sh); in the product every
// RC_KNOB(...) is a null pointer at compile time.  (Switches that tests use and that cannot produce a wrong record - RC_REDUCE_GUARDED_LOADS,
// RC_ZSTD_LITS_ALWAYS / RC_ZSTD_SEQ_ALWAYS - and plain tuning / measurement settings - RC_DEVICE, RC_DECODE_THREADS,
// RC_READ_THREADS, RC_PROFILE_ALL_STAGES, RC_READ_TIMING, RC_READ_SERIAL - stay ordinary getenv calls.)
#ifdef RC_DEV_KNOBS
#include <cstdlib>
#define RC_KNOB(name) getenv(name)
#else
#define RC_KNOB(name) (static_cast<const char *>(nullptr))
#endif

namespace rc {

// ---- geometry -----------------------------------------------------------------------------------
// A frame of N = nx*ny uint16 pixels is cut, in row-major (linear) order, into tiles of TILE_PX pixels, one tile per
// WAVEFRONT: a "group" is 512 consecutive pixels = one 16-byte load per lane (64 lanes x 8 px), a tile is R groups.
// 8 pixels per lane == exactly one bitmap byte, so a tile is TILE_BM = 512 bitmap bytes == one LZ4 / zstd block.
// Everything a tile produces (bitmap bytes, residuals, encoded block) is wave-local: no barrier, no cross-wave LDS.
// A 256-thread workgroup covers WAVES consecutive tiles and keeps them for several consecutive frames.
constexpr int WG = 256;                    // threads per workgroup (4 wavefronts of 64)
constexpr int WAVES = WG / 64;
constexpr int R = 8;                       // 16-byte loads per lane per frame-tile
constexpr int GROUP_PX = 64 * 8;           // 512
constexpr int TILE_PX = R * GROUP_PX;      // 4096 pixels = 8 KiB of uint16
constexpr int TILE_BM = TILE_PX / 8;       // 512 bitmap bytes per tile
#ifndef RC_SLOT_PX
#define RC_SLOT_PX TILE_PX
#endif
constexpr int SLOT_PX = RC_SLOT_PX;         // uint16 values between two tiles' residual slots (experiments: a denser stride; the product: one slot = one tile's worst case)
constexpr int BLK_SLOT = TILE_BM + 128;    // per-tile scratch slot for an encoded block (4-byte size word + payload), 5 x 128-byte lines

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

// ---- wavefront primitives (64 lanes) ------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_zero(uint32_t x)
{
    // lanes whose DPP source is invalid, or whose row is masked off, receive 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true);
}

// inclusive add-scan across the 64 lanes: 4 row_shr steps inside each row of 16, then two row broadcasts
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
    x += dpp_zero<0x111>(x);        // row_shr:1
    x += dpp_zero<0x112>(x);        // row_shr:2
    x += dpp_zero<0x114>(x);        // row_shr:4
    x += dpp_zero<0x118>(x);        // row_shr:8
    x += dpp_zero<0x142, 0xA>(x);   // row_bcast:15 -> rows 1 and 3
    x += dpp_zero<0x143, 0xC>(x);   // row_bcast:31 -> rows 2 and 3
    return x;
}

// value of lane-1 (lane 0 gets 0) / lane+1 (lane 63 gets 0): one DPP move each
__device__ __forceinline__ uint32_t wave_prev(uint32_t x) { return dpp_zero<0x138>(x); }  // wave_shr:1
__device__ __forceinline__ uint32_t wave_next(uint32_t x) { return dpp_zero<0x130>(x); }  // wave_shl:1

__device__ __forceinline__ uint32_t wave_last(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }

__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t x)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = __shfl_up(x, d);
        if (lane_id() >= d) x += y;
    }
    return x;
}

// block-wide exclusive scan of one uint32 per thread (WG threads); returns exclusive prefix, *total = sum.
// sm must hold WAVES+1 uint32.  Contains two barriers.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sm, uint32_t *total)
{
    const int w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane_id() == 63) sm[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
        uint32_t t = sm[i];
        if (i < w) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---- status word written by the layout kernel, read by every later kernel of the batch -----------------
struct BatchStatus {
    int32_t code;       // rc_status
    uint32_t frame;     // first offending frame
    uint64_t total;     // total record bytes of the batch
};
// "the first batch that failed since the last sync" (Scratch::first_err->total; 0 = none): ~(batch << 40 | -code << 32 | frame) - the earliest
// batch has the LARGEST key, so concurrent second stages settle it with one atomicMax
__host__ __device__ inline unsigned long long first_err_key(uint32_t batch_seq, int32_t code, uint32_t frame)
{
    return ~(((unsigned long long)(batch_seq & 0xFFFFFFu) << 40) | ((unsigned long long)((uint32_t)(-code) & 0xFFu) << 32) | frame);
}
inline BatchStatus first_err_decode(unsigned long long key)
{
    const unsigned long long k = ~key;
    return BatchStatus{-(int32_t)((k >> 32) & 0xFFu), (uint32_t)k, k >> 40};
}

}  // namespace rc
