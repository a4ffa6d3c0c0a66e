// rc_reduce.hip - threshold/binarise/compact kernel, scans, record layout and record assembly (gfx950).
//
// Reference stages restated here (paths relative to the reference repo):
//   A1 thr = dark + eps                         pyrecode/recode_writer.py:126-137
//   A2 binary = frame > thr                     pyrecode/recode_writer.py:437
//   A3 pix = frame[binary] - thr[binary]        pyrecode/recode_writer.py:440
//   A4 LSB-first bitmap                         pyrecode/recode_writer.py:622-634
//   A5 LSB-first d-bit pack                     pyrecode/recode_writer.py:637-652
//   A7 record assembly                          pyrecode/recode_writer.py:485-494,518-525,546-550,559-574
#include "rc_launch.h"

namespace rc {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b)
{
    uint32_t r;  // per 16-bit half: max(a - b, 0)  ==  (a > b) ? a - b : 0
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- A1 -------------------------------------------------------------------------------------------------
__global__ void k_threshold(const uint16_t *__restrict__ dark, uint32_t eps16, uint64_t N, uint16_t *__restrict__ thr)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < N; i += stride) thr[i] = (uint16_t)(dark[i] + eps16);
}

void launch_threshold(const uint16_t *dark, int64_t eps, uint64_t N, uint16_t *thr, hipStream_t s)
{
    uint32_t blocks = (uint32_t)((N + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_threshold, dim3(blocks), dim3(256), 0, s, dark, (uint32_t)((uint64_t)eps & 0xFFFF), N, thr);
}

// ---- A2+A3+A4: one pass over the frames ------------------------------------------------------------------
// Load 8 pixels (one bitmap byte) for this lane at pixel index px0; out-of-frame pixels read as `fill`.
template <bool ALIGNED, bool STREAM>
__device__ __forceinline__ u32x4 load8(const uint16_t *__restrict__ base, uint64_t px0, uint64_t N, uint16_t fill)
{
    if (ALIGNED) {
        if (px0 < N) {
            // frames are read exactly once (nontemporal); the threshold tile is shared by other workgroups (cached)
            const u32x4 *p = reinterpret_cast<const u32x4 *>(base + px0);
            return STREAM ? __builtin_nontemporal_load(p) : *p;
        }
        const uint32_t f2 = fill | ((uint32_t)fill << 16);
        return u32x4{f2, f2, f2, f2};
    } else {
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t k = px0 + 2 * j;
            const uint32_t lo = k < N ? base[k] : fill;
            const uint32_t hi = k + 1 < N ? base[k + 1] : fill;
            v[j] = lo | (hi << 16);
        }
        return v;
    }
}

// Workgroup `id` owns tile position id / ngroups for frames [g*BZ, g*BZ+BZ) with g = id % ngroups: workgroups that
// share a threshold tile are adjacent in dispatch order, so the tile is fetched from HBM once per batch and
// served from L2 / Infinity Cache to the other frame groups.  The threshold stays in registers for BZ frames.
template <int BZ, bool ALIGNED, bool LEVEL1>
__global__ __launch_bounds__(WG) void k_reduce_tiles(const uint16_t *__restrict__ frames,
                                                       const uint16_t *__restrict__ thr, uint64_t N, uint32_t ntiles,
                                                       uint32_t B, uint32_t ngroups, uint8_t *__restrict__ bitmap,
                                                       uint64_t nb_stride, uint16_t *__restrict__ pix_slots,
                                                       uint32_t *__restrict__ tile_cnt)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_bm[2][TILE_BM];
    __shared__ uint32_t s_wtot[2][WAVES];

    const uint32_t tile = blockIdx.x / ngroups;
    const uint32_t grp = blockIdx.x % ngroups;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const uint64_t tile_px0 = (uint64_t)tile * TILE_PX;
    const uint64_t lane_px0 = tile_px0 + (uint64_t)(w * R) * GROUP_PX + (uint64_t)lane * 8;

    u32x4 t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = load8<ALIGNED, false>(thr, lane_px0 + (uint64_t)r * GROUP_PX, N, 0xFFFF);

    const uint32_t f0 = grp * BZ;
    u32x4 x[R];
    if (f0 < B) {
        const uint16_t *fr = frames + (uint64_t)f0 * N;
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = load8<ALIGNED, true>(fr, lane_px0 + (uint64_t)r * GROUP_PX, N, 0);
    }

#pragma unroll 1
    for (int z = 0; z < BZ; ++z) {
        const uint32_t f = f0 + z;
        if (f >= B) break;
        const int buf = z & 1;

        // residuals (saturating subtract) and the 8-bit mask of this lane's 8 pixels, per group
        uint32_t m8[R];
        u32x4 res[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u32x4 d;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = pk_sub_sat_u16(x[r][j], t[r][j]);
            res[r] = d;
            const uint32_t one = 0x00010001u;
            const uint32_t M = pk_min_u16(d[0], one) | (pk_min_u16(d[1], one) << 2) | (pk_min_u16(d[2], one) << 4) |
                               (pk_min_u16(d[3], one) << 6);  // pixel 2j -> bit 2j, pixel 2j+1 -> bit 16+2j
            m8[r] = (M | (M >> 15)) & 0xFFu;
        }
        // prefetch the next frame of this tile while the current one is compacted
        if (z + 1 < BZ && f + 1 < B) {
            const uint16_t *fr = frames + (uint64_t)(f + 1) * N;
#pragma unroll
            for (int r = 0; r < R; ++r) x[r] = load8<ALIGNED, true>(fr, lane_px0 + (uint64_t)r * GROUP_PX, N, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) s_bm[buf][(w * R + r) * 64 + lane] = (uint8_t)m8[r];

        uint32_t excl[R];
        uint32_t wave_total = 0;
        if (LEVEL1) {
            // exclusive prefix of the per-lane popcounts in (group, lane) order: three groups per packed scan
            // (each field <= 512 needs 10 bits)
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += 3) {
                uint32_t pk = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (r0 + k < R) pk |= (uint32_t)__builtin_popcount(m8[r0 + k]) << (10 * k);
                const uint32_t inc = wave_incl_scan(pk);
                const uint32_t tot = wave_last(inc);
                const uint32_t exc = inc - pk;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (r0 + k < R) {
                        excl[r0 + k] = wave_total + ((exc >> (10 * k)) & 0x3FFu);
                        wave_total += (tot >> (10 * k)) & 0x3FFu;
                    }
            }
            if (lane == 0) s_wtot[buf][w] = wave_total;
        }
        __syncthreads();

        // bitmap: 8 contiguous bytes per thread, coalesced
        {
            const u32x2 b = *reinterpret_cast<const u32x2 *>(&s_bm[buf][threadIdx.x * 8]);
            uint8_t *dst = bitmap + (uint64_t)f * nb_stride + (uint64_t)tile * TILE_BM + threadIdx.x * 8;
            *reinterpret_cast<u32x2 *>(dst) = b;
        }
        if (LEVEL1) {
            uint32_t base = 0, total = 0;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) {
                const uint32_t v = s_wtot[buf][i];
                if (i < w) base += v;
                total += v;
            }
            if (threadIdx.x == 0) tile_cnt[(uint64_t)f * ntiles + tile] = total;
            uint16_t *slot = pix_slots + ((uint64_t)f * ntiles + tile) * TILE_PX + base;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t m = m8[r];
                if (m) {
                    uint32_t o = excl[r];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (m & (1u << i)) {
                            const uint32_t d = res[r][i >> 1];
                            slot[o++] = (uint16_t)((i & 1) ? (d >> 16) : (d & 0xFFFFu));
                        }
                    }
                }
            }
        }
    }
}

void launch_reduce(const Scratch &sc, const uint16_t *frames, uint32_t B, bool level1, hipStream_t s)
{
    constexpr int BZ = 4;
    const uint32_t ngroups = (B + BZ - 1) / BZ;
    const dim3 grid(sc.ntiles * ngroups), block(WG);
    const bool aligned = (sc.N % 8 == 0) && ((reinterpret_cast<uintptr_t>(frames) & 15) == 0);
#define RC_LAUNCH(AL, L1)                                                                                          \
    hipLaunchKernelGGL((k_reduce_tiles<BZ, AL, L1>), grid, block, 0, s, frames, sc.thr, sc.N, sc.ntiles, B, ngroups, \
                       sc.bitmap, sc.nb_stride, sc.pix_slots, sc.tile_cnt)
    if (aligned) {
        if (level1) RC_LAUNCH(true, true); else RC_LAUNCH(true, false);
    } else {
        if (level1) RC_LAUNCH(false, true); else RC_LAUNCH(false, false);
    }
#undef RC_LAUNCH
}

// ---- per-frame scans over tiles ---------------------------------------------------------------------------
// One workgroup per frame: exclusive prefix of in[f][0..ntiles) -> off, row total -> total[f]; optionally
// next[f][t] = smallest t' > t with in[f][t'] > 0 (ntiles if none).
template <bool WITH_NEXT>
__global__ __launch_bounds__(WG) void k_scan_rows(const uint32_t *__restrict__ in, uint32_t *__restrict__ off,
                                                    uint32_t *__restrict__ total, uint32_t *__restrict__ next,
                                                    uint32_t ntiles, const BatchStatus *__restrict__ st)
{
    __shared__ uint32_t sm[WAVES + 1];
    __shared__ uint32_t s_carry;
    const uint32_t f = blockIdx.x;
    const uint32_t *row = in + (uint64_t)f * ntiles;
    uint32_t *orow = off + (uint64_t)f * ntiles;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += WG) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < ntiles ? row[t] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, sm, &tot);
        if (t < ntiles) orow[t] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) total[f] = carry;
    if (WITH_NEXT) {
        // suffix pass, chunks from the end: next non-empty tile index
        uint32_t *nrow = next + (uint64_t)f * ntiles;
        if (threadIdx.x == 0) s_carry = ntiles;
        __syncthreads();
        const uint32_t nchunks = (ntiles + WG - 1) / WG;
        for (uint32_t c = nchunks; c-- > 0;) {
            const uint32_t t = c * WG + threadIdx.x;
            const uint32_t v = t < ntiles ? row[t] : 0;
            // candidate = own index if non-empty else "infinite"; suffix-min over lanes to the right (exclusive)
            uint32_t cand = v ? t : 0xFFFFFFFFu;
            // inclusive suffix-min inside the wave via shuffles
            uint32_t m = cand;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_down(m, d);
                if (lane_id() + d < 64) m = min(m, y);
            }
            const int w = threadIdx.x >> 6;
            if (lane_id() == 0) sm[w] = m;  // min of the whole wave
            __syncthreads();
            uint32_t right = s_carry;  // min over all later chunks
#pragma unroll
            for (int i = WAVES - 1; i >= 0; --i)
                if (i > w) right = min(right, sm[i]);
            uint32_t excl_in_wave = __shfl_down(m, 1);
            if (lane_id() == 63) excl_in_wave = 0xFFFFFFFFu;
            const uint32_t nx = min(excl_in_wave, right);
            if (t < ntiles) nrow[t] = nx;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t all = s_carry;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) all = min(all, sm[i]);
                s_carry = all;
            }
            __syncthreads();
        }
    }
    (void)st;
}

void launch_scan_counts(const Scratch &sc, uint32_t B, hipStream_t s)
{
    hipLaunchKernelGGL((k_scan_rows<true>), dim3(B), dim3(WG), 0, s, sc.tile_cnt, sc.tile_off, sc.frame_nnz,
                       sc.tile_next, sc.ntiles, sc.status);
}
void launch_scan_blocks(const Scratch &sc, uint32_t B, hipStream_t s)
{
    hipLaunchKernelGGL((k_scan_rows<false>), dim3(B), dim3(WG), 0, s, sc.blk_size, sc.blk_off, sc.frame_cbytes,
                       (uint32_t *)nullptr, sc.ntiles, sc.status);
}

// ---- record layout: sizes, offsets, metadata, status ------------------------------------------------------------
constexpr uint32_t LZ4F_HDR = 7;                // magic + FLG + BD + HC
constexpr uint32_t LZ4F_END = 4;                // EndMark
constexpr uint32_t LZ4F_MAXBLK_SHIFT = 22;      // 4 MiB stored chunks for the pixel stream

__host__ __device__ inline uint32_t packed_bytes(uint32_t nnz, uint32_t depth)
{
    return depth == 16 ? nnz * 2u : (uint32_t)(((uint64_t)nnz * depth + 7) >> 3);
}
__host__ __device__ inline uint32_t lz4f_stored_size(uint32_t n)
{
    const uint32_t chunks = (n + (1u << LZ4F_MAXBLK_SHIFT) - 1) >> LZ4F_MAXBLK_SHIFT;
    return LZ4F_HDR + n + 4 * chunks + LZ4F_END;
}

__global__ __launch_bounds__(WG) void k_layout(const uint32_t *__restrict__ frame_nnz,
                                                 const uint32_t *__restrict__ frame_cbytes, RecordParams rp, uint64_t nb,
                                                 uint32_t B, uint64_t out_cap, uint64_t *__restrict__ rec_off,
                                                 uint32_t *__restrict__ md, BatchStatus *__restrict__ st)
{
    __shared__ uint64_t s_part[WG];
    __shared__ uint32_t s_bad;
    if (threadIdx.x == 0) s_bad = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t per = (B + WG - 1) / WG;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = min(lo + per, B);
    uint64_t sum = 0;
    for (uint32_t f = lo; f < hi; ++f) {
        const uint32_t nnz = rp.level == 1 ? frame_nnz[f] : 0;
        const uint32_t npk = rp.level == 1 ? packed_bytes(nnz, rp.depth) : 0;
        uint64_t sz;
        uint32_t m0 = 0, m1 = 0, m2 = 0;
        if (rp.emit == 0) {
            if (rp.level == 1) { sz = 8 + nb + npk; m0 = npk; }
            else sz = 4 + nb;
        } else {  // LZ4 frames (emit == 2)
            const uint32_t cb = LZ4F_HDR + frame_cbytes[f] + LZ4F_END;
            if (rp.level == 1) {
                const uint32_t cp = lz4f_stored_size(npk);
                sz = 16 + (uint64_t)cb + cp; m0 = cb; m1 = cp; m2 = npk;
            } else { sz = 8 + (uint64_t)cb; m0 = cb; }
        }
        md[3 * f] = m0; md[3 * f + 1] = m1; md[3 * f + 2] = m2;
        if (sz > rp.frame_bytes) atomicMin(&s_bad, f);
        rec_off[f + 1] = sz;  // sizes first; turned into offsets below
        sum += sz;
    }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t i = 0; i < threadIdx.x; ++i) base += s_part[i];
    for (uint32_t f = lo; f < hi; ++f) {
        const uint64_t sz = rec_off[f + 1];
        base += sz;
        rec_off[f + 1] = base;
    }
    if (threadIdx.x == 0) rec_off[0] = 0;
    __syncthreads();
    if (threadIdx.x == WG - 1) {
        uint64_t total = 0;
        for (uint32_t i = 0; i < WG; ++i) total += s_part[i];
        st->total = total;
        if (s_bad != 0xFFFFFFFFu) { st->code = -5; st->frame = s_bad; }        // RC_ERR_RECORD_TOO_LARGE
        else if (total > out_cap) { st->code = -2; st->frame = 0; }            // RC_ERR_OUT_TOO_SMALL
    }
}

void launch_layout(const Scratch &sc, const RecordParams &rp, uint32_t B, uint64_t out_cap, uint64_t *rec_off,
                   uint32_t *md, hipStream_t s)
{
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(WG), 0, s, sc.frame_nnz, sc.frame_cbytes, rp, sc.nb, B, out_cap,
                       rec_off, md, sc.status);
}

// ---- record assembly ------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_u32_le(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

// byte copy by one wavefront, arbitrary alignment on both sides
__device__ __forceinline__ void wave_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t n)
{
    for (uint32_t i = lane_id(); i < n; i += 64) dst[i] = src[i];
}

struct PixSrc {
    const uint16_t *slots;       // frame base: [ntiles][TILE_PX]
    const uint32_t *cnt, *next;  // frame rows
    uint32_t ntiles, t, P, c, nnz;
};
// value with frame-level index v >= P (this tile's first value), 0 beyond the frame's last value
__device__ __forceinline__ uint32_t pix_fetch(const PixSrc &s, uint32_t v)
{
    if (v >= s.nnz) return 0;
    uint32_t idx = v - s.P;
    if (idx < s.c) return s.slots[(uint64_t)s.t * TILE_PX + idx];
    idx -= s.c;
    uint32_t tt = s.next[s.t];
    while (tt < s.ntiles) {
        const uint32_t cc = s.cnt[tt];
        if (idx < cc) return s.slots[(uint64_t)tt * TILE_PX + idx];
        idx -= cc;
        tt = s.next[tt];
    }
    return 0;
}

// position of packed-pixel byte b inside the pixel LZ4 frame (stored chunks of 4 MiB)
__device__ __forceinline__ uint64_t lz4f_stored_pos(uint64_t b) { return LZ4F_HDR + 4 * ((b >> LZ4F_MAXBLK_SHIFT) + 1) + b; }

// One wavefront per (tile, frame): copies the tile's encoded bitmap block (or raw bitmap bytes) to its place in the
// record and bit-packs the tile's residuals straight into the record; wave 0 of tile 0 writes the fixed fields.
__global__ __launch_bounds__(WG) void k_assemble(Scratch sc, RecordParams rp, uint32_t B, uint8_t *__restrict__ out,
                                                   const uint64_t *__restrict__ rec_off, uint32_t lz4f_hdr_bitmap,
                                                   uint32_t lz4f_hdr_pix)
{
    if (sc.status->code != 0) return;
    const uint32_t f = blockIdx.y;
    const uint32_t t = blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int lane = lane_id();
    if (t >= sc.ntiles) return;
    uint8_t *rec = out + rec_off[f];
    const uint64_t frow = (uint64_t)f * sc.ntiles;
    const uint32_t nnz = rp.level == 1 ? sc.frame_nnz[f] : 0;
    const uint32_t npk = rp.level == 1 ? packed_bytes(nnz, rp.depth) : 0;

    uint64_t bitmap_pos, pix_pos;  // record offsets of the bitmap stream and of packed-pixel byte 0's container
    uint32_t cb = 0;
    if (rp.emit == 0) {
        bitmap_pos = rp.level == 1 ? 8 : 4;
        pix_pos = bitmap_pos + sc.nb;
    } else {
        cb = LZ4F_HDR + sc.frame_cbytes[f] + LZ4F_END;
        bitmap_pos = rp.level == 1 ? 16 : 8;
        pix_pos = bitmap_pos + cb;
    }

    // fixed fields
    if (t == 0 && lane == 0) {
        store_u32_le(rec, rp.first_frame_id + f);
        if (rp.emit == 0) {
            if (rp.level == 1) store_u32_le(rec + 4, npk);
        } else {
            store_u32_le(rec + 4, cb);
            uint8_t *bf = rec + bitmap_pos;
            store_u32_le(bf, 0x184D2204u);
            bf[4] = (uint8_t)(lz4f_hdr_bitmap & 0xFF); bf[5] = (uint8_t)((lz4f_hdr_bitmap >> 8) & 0xFF);
            bf[6] = (uint8_t)((lz4f_hdr_bitmap >> 16) & 0xFF);
            store_u32_le(bf + cb - LZ4F_END, 0);
            if (rp.level == 1) {
                const uint32_t cp = lz4f_stored_size(npk);
                store_u32_le(rec + 8, cp);
                store_u32_le(rec + 12, npk);
                uint8_t *pf = rec + pix_pos;
                store_u32_le(pf, 0x184D2204u);
                pf[4] = (uint8_t)(lz4f_hdr_pix & 0xFF); pf[5] = (uint8_t)((lz4f_hdr_pix >> 8) & 0xFF);
                pf[6] = (uint8_t)((lz4f_hdr_pix >> 16) & 0xFF);
                const uint32_t chunk = 1u << LZ4F_MAXBLK_SHIFT;
                for (uint32_t k = 0, o = 0; o < npk; ++k, o += chunk) {
                    const uint32_t len = min(chunk, npk - o);
                    store_u32_le(pf + LZ4F_HDR + (uint64_t)k * (chunk + 4), len | 0x80000000u);
                }
                store_u32_le(pf + cp - LZ4F_END, 0);
            }
        }
    }

    // bitmap stream
    if (rp.emit == 0) {
        const uint64_t b0 = (uint64_t)t * TILE_BM;
        const uint32_t n = (uint32_t)min((uint64_t)TILE_BM, sc.nb - b0);
        wave_copy(rec + bitmap_pos + b0, sc.bitmap + (uint64_t)f * sc.nb_stride + b0, n);
    } else {
        const uint32_t n = sc.blk_size[frow + t];
        wave_copy(rec + bitmap_pos + LZ4F_HDR + sc.blk_off[frow + t], sc.blk_slots + (frow + t) * BLK_SLOT, n);
    }

    // packed residuals owned by this tile
    if (rp.level != 1) return;
    const uint32_t c = sc.tile_cnt[frow + t];
    if (c == 0) return;
    const uint32_t P = sc.tile_off[frow + t];
    uint8_t *pdst = rec + pix_pos;
    const uint32_t d = rp.depth;
    if (d == 16) {
        const uint16_t *src = sc.pix_slots + (frow + t) * TILE_PX;
        for (uint32_t i = lane; i < c; i += 64) {
            const uint32_t v = src[i];
            const uint64_t b = 2ull * (P + i);
            if (rp.emit == 0) { pdst[b] = (uint8_t)v; pdst[b + 1] = (uint8_t)(v >> 8); }
            else { pdst[lz4f_stored_pos(b)] = (uint8_t)v; pdst[lz4f_stored_pos(b + 1)] = (uint8_t)(v >> 8); }
        }
        return;
    }
    PixSrc ps{sc.pix_slots + frow * TILE_PX, sc.tile_cnt + frow, sc.tile_next + frow, sc.ntiles, t, P, c, nnz};
    const uint32_t dmask = (1u << d) - 1;
    const uint64_t b_lo = ((uint64_t)P * d + 7) >> 3;
    const uint64_t b_hi = ((uint64_t)(P + c) * d + 7) >> 3;  // exclusive: bytes whose first bit lies in this tile's values
    for (uint64_t b = b_lo + lane; b < b_hi; b += 64) {
        const uint64_t bit0 = b * 8;
        uint32_t v = (uint32_t)(bit0 / d);
        const uint32_t o = (uint32_t)(bit0 - (uint64_t)v * d);
        uint32_t acc = (pix_fetch(ps, v) & dmask) >> o;
        uint32_t filled = d - o;
        while (filled < 8) {
            ++v;
            acc |= (pix_fetch(ps, v) & dmask) << filled;
            filled += d;
        }
        pdst[rp.emit == 0 ? b : lz4f_stored_pos(b)] = (uint8_t)acc;
    }
}

// xxHash32 of the two descriptor bytes -> LZ4 frame header checksum byte (lz4_Frame_format.md, "HC")
static uint32_t xxh32_small(const uint8_t *p, size_t n)
{
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    (void)P1; (void)P4;
    uint32_t h = 0 /*seed*/ + P5 + (uint32_t)n;
    for (size_t i = 0; i < n; ++i) {
        h += p[i] * P5;
        h = ((h << 11) | (h >> 21)) * P1;
    }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
uint32_t lz4f_descriptor(uint8_t bd)
{
    const uint8_t desc[2] = {0x60 /* version 01, block-independent, no checksums, no content size */, bd};
    const uint8_t hc = (uint8_t)((xxh32_small(desc, 2) >> 8) & 0xFF);
    return desc[0] | ((uint32_t)desc[1] << 8) | ((uint32_t)hc << 16);
}

void launch_assemble(const Scratch &sc, const RecordParams &rp, uint32_t B, uint8_t *out, const uint64_t *rec_off,
                     hipStream_t s)
{
    static const uint32_t hdr_bitmap = lz4f_descriptor(0x40);  // 64 KiB max block (blocks are <= 2 KiB)
    static const uint32_t hdr_pix = lz4f_descriptor(0x70);     // 4 MiB max block (stored chunks)
    const dim3 grid((sc.ntiles + WAVES - 1) / WAVES, B), block(WG);
    hipLaunchKernelGGL(k_assemble, grid, block, 0, s, sc, rp, B, out, rec_off, hdr_bitmap, hdr_pix);
}

}  // namespace rc
