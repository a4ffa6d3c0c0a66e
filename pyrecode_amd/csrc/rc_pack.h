// rc_pack.h - the tile-local d-bit pack in a wavefront's LDS stage (shared by the reduce kernel and the level-2 emit).
// Reference: the LSB-first concatenation of the values' low d bits, pyrecode/recode_writer.py:637-652 (A5).
#pragma once
#include "rc_device.h"

namespace rc {

// A5 inside the tile: the cnt compacted values (uint16, in the wave's LDS stage) become the tile-local LSB-first stream of
// their low d bits, IN PLACE: output dword w needs values >= 32w/d >= 2w, which lie at or behind byte 4w, and all lanes
// of a step read before any of them writes.  The rest of the last 128-byte line is zeroed (k_gather reads whole 16-byte pieces).
__device__ __forceinline__ void pack_stage(uint16_t *pix, uint32_t cnt, uint32_t d)
{
    const int lane = lane_id();
    const uint32_t nbits = cnt * d;
    const uint32_t ndw = (((nbits + 31) >> 5) + 31u) & ~31u;
    if (d == 12) {
        // the common detector depth: 8 values = 3 dwords per lane and step, one 16-byte LDS read, no inner loop.  In place is
        // safe for the same reason as below (12g <= 16g: a group's output lies at or in front of its input, and a step's
        // reads all happen before its writes); the padding up to the line boundary is zeroed behind.
        uint32_t *out = reinterpret_cast<uint32_t *>(pix);
        const uint32_t ngrp = (cnt + 7) >> 3;
        for (uint32_t g0 = 0; g0 < ngrp; g0 += 64) {
            const uint32_t g = g0 + lane;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (g < ngrp) v = *reinterpret_cast<const u32x4 *>(pix + 8 * g);
            uint32_t x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t h = (j & 1) ? (v[j >> 1] >> 16) : (v[j >> 1] & 0xFFFFu);
                x[j] = 8 * g + j < cnt ? (h & 0xFFFu) : 0u;
            }
            const uint32_t w0 = x[0] | (x[1] << 12) | (x[2] << 24);
            const uint32_t w1 = (x[2] >> 8) | (x[3] << 4) | (x[4] << 16) | (x[5] << 28);
            const uint32_t w2 = (x[5] >> 4) | (x[6] << 8) | (x[7] << 20);
            __builtin_amdgcn_wave_barrier();
            if (g < ngrp) { out[3 * g] = w0; out[3 * g + 1] = w1; out[3 * g + 2] = w2; }
            __builtin_amdgcn_wave_barrier();
        }
        for (uint32_t w = 3 * ngrp + lane; w < ndw; w += 64) out[w] = 0;
        __builtin_amdgcn_wave_barrier();
        return;
    }
    const uint32_t inv = 0xFFFFFFFFu / d + 1u;  // floor(n / d) = umulhi(n, inv) for n * d < 2^32  (d = 1: the constant wraps to 0 - taken apart below)
    const uint32_t dmask = (1u << d) - 1u;
    uint32_t *out = reinterpret_cast<uint32_t *>(pix);
    for (uint32_t w0 = 0; w0 < ndw; w0 += 64) {
        const uint32_t w = w0 + lane;
        uint32_t v = d == 1 ? 32u * w : __umulhi(32u * w, inv);
        const uint32_t o = 32u * w - v * d;
        uint64_t acc = 0;
        uint32_t filled = 0;
        if (v < cnt) { acc = (pix[v] & dmask) >> o; filled = d - o; ++v; }
        while (filled < 32 && v < cnt) {
            acc |= (uint64_t)(pix[v] & dmask) << filled;
            filled += d;
            ++v;
        }
        __builtin_amdgcn_wave_barrier();
        if (w < ndw) out[w] = (uint32_t)acc;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace rc
