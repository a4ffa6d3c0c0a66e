// rc_deflate.hip - the Adler-32 trailers of a batch's zlib streams (compression_scheme 0 on the device, gfx950).
//
// Reference: `zlib.compress(data, level)` on the packed binary map and on the packed residuals (pyrecode/recode_compressors.py:84-85,
// recode_writer.py:503-511).  A zlib stream ends with the Adler-32 of the UNcompressed bytes (RFC 1950), big-endian:
//   s1 = 1 + sum b_i,  s2 = sum of the running s1 = n + sum (n - i) b_i   (mod 65521)   =>   with A = sum b_i, W = sum i b_i:
//   s1 = 1 + A,  s2 = n (1 + A) - W.
// The map's A and W are sums of per-tile partials the reduce kernel left (rc_deflate_block.h::deflate_adler_word; the map itself is
// never written).  The residual stream exists only in the record (k_gather concatenates the tiles' d-bit streams at bit granularity), so
// its sums are taken from there: every workgroup reads a slice of the stored blocks' payload the gather has just written (0.3 MB a frame
// at 1 %: 1 % of what the step reads) - 16 stream bytes per lane and step, v_sad_u8 / v_dot4 for the two sums.
// One launch: grid (segments, frames); the segments' sums meet in the frame's accumulators (zeroed by k_layout), the last workgroup
// to arrive writes both trailers.
#include "rc_launch.h"
#include "rc_record.h"
#include "rc_deflate_block.h"

namespace rc {

constexpr int ZT = 256;   // threads per workgroup: small enough to start next to a running reduce kernel (see k_scan_frames)

typedef u32x4 u32x4_b __attribute__((aligned(1)));

__device__ __forceinline__ void store_u32_be(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}
__device__ __forceinline__ uint32_t adler_from_sums(uint64_t n, uint32_t A, uint32_t W)   // A, W: any representatives mod 65521
{
    const uint64_t a = A % ADLER_P, w = W % ADLER_P, nm = n % ADLER_P;
    const uint32_t s1 = (uint32_t)((1 + a) % ADLER_P);
    const uint32_t s2 = (uint32_t)((nm * (1 + a) + ADLER_P - w) % ADLER_P);
    return (s2 << 16) | s1;
}

__global__ __launch_bounds__(ZT) void k_zlib_trailers(Scratch sc, RecordParams rp, uint8_t *__restrict__ out, const uint64_t *__restrict__ rec_off, uint32_t nseg)
{
    if (sc.status->code != 0) return;   // (a failed batch: k_gather has noted it, the records are undefined)
    __shared__ uint32_t s_red[4][ZT / 64];
    __shared__ uint32_t s_lastwg;
    const uint32_t seg = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const FrameFmt ff = frame_fmt(rp.emit);
    const uint64_t frow = (uint64_t)f * sc.ntiles;
    uint8_t *rec = out + rec_off[f];
    const uint32_t cb = bitmap_hdr(ff, rp.emit, sc.ntiles) + sc.frame_cbytes[f] + ff.end;
    const uint32_t bitmap_pos = rp.level == 1 ? 16u : 8u;
    const uint32_t nnz = rp.level == 1 ? sc.frame_nnz[f] : 0u;
    const uint32_t npk = rp.level == 1 ? packed_bytes(nnz, rp.depth) : 0u;
    const uint8_t *pf = rec + bitmap_pos + cb;     // the residual stream's zlib stream

    // ---- the map: this segment's tiles ---------------------------------------------------------------------------------------------
    uint32_t mA = 0, mW = 0;
    {
        const uint32_t per = (sc.ntiles + nseg - 1) / nseg, lo = seg * per, hi = min(lo + per, sc.ntiles);
        for (uint32_t t = lo + tid; t < hi; t += ZT) {
            const uint32_t w = sc.blk_aux[frow + t];
            mA += w & 0xFFFFu;
            mW += w >> 16;
        }
    }
    // ---- the residual stream: this segment's 16-byte units -----------------------------------------------------------------------------
    uint32_t pA = 0;
    uint64_t pW = 0;
    if (npk) {
        const uint32_t units = (npk + 15) >> 4, per = (units + nseg - 1) / nseg, lo = seg * per, hi = min(lo + per, units);
        for (uint32_t u = lo + tid; u < hi; u += ZT) {
            const uint32_t b = 16u * u;                                     // (a unit never straddles a stored block's header: blocks are 32 KiB)
            const uint8_t *p = pf + stored_pos(ff, b);
            u32x4 x;
            if (b + 16 <= npk) x = *reinterpret_cast<const u32x4_b *>(p);
            else {
                x = u32x4{0u, 0u, 0u, 0u};
                for (uint32_t i = 0; b + i < npk; ++i) x[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3u));
            }
            uint32_t a = 0, q = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a = __builtin_amdgcn_sad_u8(x[k], 0u, a);
                q = __builtin_amdgcn_udot4(x[k], 0x03020100u + 0x04040404u * (uint32_t)k, q, false);
            }
            pA += a;
            pW += (uint64_t)b * a + q;
        }
    }
    // ---- workgroup sums (every term reduced mod 65521 first: the frame's accumulators hold plain sums of such terms) ----------------------
    uint32_t v[4] = {mA, mW, pA % ADLER_P, (uint32_t)(pW % ADLER_P)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t tot = wave_last(wave_incl_scan(v[k]));
        if (lane_id() == 0) s_red[k][tid >> 6] = tot;
    }
    __syncthreads();
    uint32_t *acc = sc.zl_acc + 8 * (uint64_t)f;
    if (tid < 4) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < ZT / 64; ++w) tot += s_red[tid][w];
        atomicAdd(&acc[tid], tot % ADLER_P);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_lastwg = atomicAdd(&acc[4], 1u) == nseg - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_lastwg || tid != 0) return;
    __threadfence();
    const uint32_t A0 = __hip_atomic_load(&acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), W0 = __hip_atomic_load(&acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t A1 = __hip_atomic_load(&acc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), W1 = __hip_atomic_load(&acc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    store_u32_be(rec + bitmap_pos + cb - 4, adler_from_sums(sc.nb, A0, W0));
    if (rp.level == 1) store_u32_be(rec + bitmap_pos + cb + stored_size(ff, npk) - 4, adler_from_sums(npk, A1, W1));
}

void launch_zlib_trailers(const Scratch &sc, const RecordParams &rp, uint32_t B, uint8_t *out, const uint64_t *rec_off, hipStream_t s)
{
    // segments per frame: enough workgroups to cover the chip's latency (the residual streams are read once), at most 64 per frame
    uint32_t nseg = sc.ntiles / 256;
    if (nseg < 1) nseg = 1;
    if (nseg > 64) nseg = 64;
    hipLaunchKernelGGL(k_zlib_trailers, dim3(nseg, B), dim3(ZT), 0, s, sc, rp, out, rec_off, nseg);
}

}  // namespace rc
