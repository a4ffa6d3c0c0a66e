// rc_gather.hip - record assembly, second form (round 5): the tiles' encoded blocks and packed residuals -> the records (gfx950).
// Reference: record layout pyrecode/recode_writer.py:485-494,518-525,546-550,559-574 (A7), d-bit concatenation :637-652 (A5).
#include <cstdlib>

#include "rc_launch.h"
#include "rc_record.h"
#include "rc_deflate_block.h"

namespace rc {

// ---- record assembly, second form (round 5): k_gather ---------------------------------------------------------------------------------
// Its predecessor, k_assemble (rounds 1-4), gave every tile a quarter-wave: 32 768 short-lived wavefronts per 64 frames of 4096^2, each with three dependent
// round trips (indices, data, stores) for 1.8 KB of payload, in four-wave workgroups that only start where a reduce workgroup of the
// NEXT batch has just retired - 85 us alone, 300 us next to the reduce kernel, and what it takes from that kernel is what the step
// loses (profiles/r04_decompose_*.log).  k_gather does the same copy with few, long-lived, one-wave workgroups that need no LDS
// and few registers, so that they live in what the reduce kernel's workgroups leave free on a CU (its 15 waves of 128 registers
// leave one SIMD a quarter empty) instead of displacing them:
//   - an ITEM is 64 consecutive tiles of one frame, a lane per tile for the bookkeeping: sizes, offsets (from the scans) and where
//     the tile's two segments - encoded block, packed residuals - come from and go to;
//   - the copy itself is PIECE-parallel: a piece is 16 bytes of a segment, the item's pieces are numbered through (exclusive scan of
//     the tiles' piece counts), lane j of round i takes piece 64 i + j, finds its tile by a binary search over the scan (six
//     ds_bpermute) and moves it: one aligned 16-byte load from the slot, one 16-byte store at the destination's byte alignment
//     (unaligned access mode), a segment's last piece as 8 + 4 + 2 + 1 bytes.  Every lane of every round carries 16 bytes, whatever
//     the tiles' sizes;
//   - residual streams of d-bit fields (d % 8 != 0, BITS) are funnel-shifted by the destination's bit phase on the way (a fifth dword
//     per piece); the stream byte a tile shares with its successor is completed from the successor's leading bits - in registers when
//     the successor is a tile of the same item, from the first tile behind the item otherwise (fetched with the item's bookkeeping),
//     through tile_next in the rare rest.
// Same bytes as k_assemble in every mode (same-box A/B and the whole suite, round 5); level 2 arrives here as tile-local packed streams too.
// what k_gather needs of the scratch set (same member names as Scratch: the kernel's argument block stays small - the whole Scratch is
// 60 scalar registers of pointers the kernel never touches, and the spills they cause sit in the item loop)
struct GatherArgs {
    const uint8_t *blk_slots, *bitmap;
    const uint16_t *pix_slots;
    const uint32_t *blk_size, *blk_off, *tile_cnt, *tile_off, *tile_next, *frame_nnz, *frame_cbytes, *frame_pbytes;
    const uint32_t *blk_aux;   // deflate: the tiles' Adler-32 partials of the map
    uint32_t *zl_acc;          // deflate: per frame {A, W of the map, A, W of the residual stream, items arrived}
    uint8_t *pixraw;
    BatchStatus *status, *first_err;
    uint64_t nb, nb_stride, pixraw_stride;
    uint32_t ntiles, blk_stride, pix_slot_bytes, comb;
};

typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef uint16_t u16_u __attribute__((aligned(1)));

__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t src_lane)   // v of lane src_lane (every lane must be active)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * src_lane), (int)v);
}

// one piece: y = the 160-bit string (x, e) >> sh, its first `valid` (1..16) bytes to p
template <bool BITS>
__device__ __forceinline__ u32x4 shift_piece(const u32x4 &x, uint32_t e, uint32_t sh)
{
    u32x4 y = x;
    if (BITS) {
        y[0] = __builtin_amdgcn_alignbit(x[1], x[0], sh); y[1] = __builtin_amdgcn_alignbit(x[2], x[1], sh);
        y[2] = __builtin_amdgcn_alignbit(x[3], x[2], sh); y[3] = __builtin_amdgcn_alignbit(e, x[3], sh);
    }
    return y;
}
__device__ __forceinline__ void put_piece(uint8_t *p, const u32x4 &y, uint32_t valid)
{
    if (valid >= 16) { *reinterpret_cast<u32x4_u *>(p) = y; return; }
    uint32_t a0 = y[0], a1 = y[1];
    if (valid & 8u) { *reinterpret_cast<u32x2_u *>(p) = u32x2{a0, a1}; p += 8; a0 = y[2]; a1 = y[3]; }
    if (valid & 4u) { *reinterpret_cast<u32_u *>(p) = a0; p += 4; a0 = a1; }
    if (valid & 2u) { *reinterpret_cast<u16_u *>(p) = (uint16_t)a0; p += 2; a0 >>= 16; }
    if (valid & 1u) *p = (uint8_t)a0;
}

// deflate (ADLER): the residual stream's Adler-32 is summed up from the bytes as they pass through the registers on their way into the record
// (rc_deflate_block.h: A = sum of the bytes, W = sum of position * byte, both mod 65521; 32-bit arithmetic: a piece of <= 16 bytes at
// stream byte sp adds (sp mod p) * a + q, a <= 4080 its byte sum, q its bytes weighted by their place in the piece)
struct AdlerAcc { uint32_t A, W; };
__device__ __forceinline__ void adler_piece(AdlerAcc &acc, const u32x4 &y, uint32_t valid, uint32_t sp)
{
    uint32_t a = 0, q = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t nb = valid > 4u * k ? min(valid - 4u * k, 4u) : 0u;
        const uint32_t v = y[k] & (nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u));
        a = __builtin_amdgcn_sad_u8(v, 0u, a);
        q = __builtin_amdgcn_udot4(v, 0x03020100u + 0x04040404u * (uint32_t)k, q, false);
    }
    acc.A += a;                                      // (a lane's pieces of one item: far below 2^32)
    acc.W = (acc.W + (sp % ADLER_P) * a + q) % ADLER_P;
}
__device__ __forceinline__ void adler_byte(AdlerAcc &acc, uint32_t byte, uint32_t sp)
{
    acc.A += byte;
    acc.W = (acc.W + (sp % ADLER_P) * byte) % ADLER_P;
}

// Where a tile's packed residuals lie in the frame's stream: coff = set pixels in front of the tile, cnt = its own, d bits each.
// The tile owns the stream bytes whose FIRST bit is one of its bits: n whole bytes from b_lo on, taken from its own packed stream at bit
// ps0; when its last owned byte is only partly its own (avail bits), that byte is completed from the successor(s) and n does not count it.
struct ResidGeom { uint32_t b_lo, n, ps0, avail; };
__device__ __forceinline__ ResidGeom resid_geom(uint32_t coff, uint32_t cnt, uint32_t d)
{
    const uint64_t dbit = (uint64_t)coff * d;
    const uint32_t nbits = cnt * d;
    const uint64_t b_lo = (dbit + 7) >> 3, b_hi = (dbit + nbits + 7) >> 3;
    ResidGeom g{(uint32_t)b_lo, (uint32_t)(b_hi - b_lo), (uint32_t)(8 * b_lo - dbit), (uint32_t)((dbit + nbits) & 7u)};
    if (g.avail && g.n) --g.n;     // (n == 0: the tile's few bits all live in a byte that an earlier tile owns)
    else g.avail = 0;
    return g;
}

// Rounds of pieces whose loads are in flight together, and the register budget: at most 64 VGPRs - two of these waves fit where one
// reduce wave (128) does, and what the reduce kernel's 15 waves leave free on a CU is one such slot.  Same-box A/B
// (profiles/r05_exp3_gather_variants.log): 4 rounds / 64 registers ahead of 4 / 74, 8 / 110 and 12 / 125 on the headline and the
// detector-like stack (+1 %); the d-bit form needs a few registers more per round, so it takes 3.
#ifndef RC_GATHER_U
#define RC_GATHER_U 4
#endif
#ifndef RC_GATHER_UB
#define RC_GATHER_UB 3
#endif
#ifndef RC_GATHER_WPE
#define RC_GATHER_WPE 8   // waves per SIMD the register allocation aims at (8: at most 64 VGPRs)
#endif
template <bool BITS, int U, bool ADLER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RC_GATHER_WPE))) void k_gather(GatherArgs sc, RecordParams rp, uint8_t *__restrict__ out, const uint64_t *__restrict__ rec_off,
                                               uint32_t lz4f_hdr_bitmap, uint32_t lz4f_hdr_pix, uint32_t batch_seq, uint32_t gpf, uint32_t nitems, uint32_t tpi)
{
    if (sc.status->code != 0) {   // (the batch's last kernel remembers the first failure across asynchronously enqueued batches)
        // ONE 64-bit key per failure, the earliest batch's the largest (rc_api.hip::rc_ctx_sync decodes it): two batches' second stages may
        // run at the same time (two chains), and a systematic failure - an output buffer too small - fails them all
        if (blockIdx.x == 0 && threadIdx.x == 0 && sc.first_err)
            atomicMax(reinterpret_cast<unsigned long long *>(&sc.first_err->total), first_err_key(batch_seq, sc.status->code, sc.status->frame));
        return;
    }
    const uint32_t lane = (uint32_t)lane_id();
    const bool flat = rp.pix_mode == 1, skip_pix = rp.pix_mode == 2;
    const bool pixp = rp.level == 1 && rp.packed_slots && !skip_pix;
    const bool plain_pos = rp.emit == 0 || flat;
    const uint32_t d = rp.depth;
    const FrameFmt ff = frame_fmt(rp.emit);
    const uint32_t bhdr = bitmap_hdr(ff, rp.emit, sc.ntiles);
    const uint8_t *pix_base = reinterpret_cast<const uint8_t *>(sc.pix_slots);
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint32_t f = item / gpf, g = item - f * gpf, t0 = tpi * g, t = t0 + lane;   // tpi tiles per item: 64, fewer for small batches
        const bool have = lane < tpi && t < sc.ntiles;
        const uint64_t frow = (uint64_t)f * sc.ntiles;
        uint8_t *rec = flat ? sc.pixraw + (uint64_t)f * sc.pixraw_stride : out + rec_off[f];
        const uint32_t nnz = rp.level == 1 ? sc.frame_nnz[f] : 0;
        const uint32_t npk = rp.level == 1 ? packed_bytes(nnz, d) : 0;
        uint64_t bitmap_pos = 0, pix_pos = 0;
        uint32_t cb = 0;
        if (!flat) {
            if (rp.emit == 0) { bitmap_pos = rp.level == 1 ? 8 : 4; pix_pos = bitmap_pos + sc.nb; }
            else { cb = bhdr + sc.frame_cbytes[f] + ff.end; bitmap_pos = rp.level == 1 ? 16 : 8; pix_pos = bitmap_pos + cb; }
        }
        if (g == 0 && lane == 0 && !flat) record_fixed_fields(sc, rp, f, rec, bitmap_pos, pix_pos, cb, npk, ff, skip_pix, lz4f_hdr_bitmap, lz4f_hdr_pix);
        uint8_t *pdst = rec + pix_pos;

        // ---- bookkeeping, a lane per tile -------------------------------------------------------------------------------------------
        // block segment: bsz bytes from the slot's start (mode 0: from the raw binary map) to rec + bdst
        // residual segment: rn whole stream bytes, source bit ps0 of the tile's packed stream onwards, to rec + rdst; source at slot + rs
        // (rs = ~0: the tile's residual slot)
        // (what the copy loop does not need - the shared byte's geometry, a slow tile's length - is not kept across it but derived again
        // behind it from cnt and a second read of tile_off: registers; `packed` carries two flags for it)
        uint32_t word = 0, bsz = 0, bdst = 0, cnt = 0, rn = 0, rdst = 0, rs = 0xFFFFFFFFu, ps0 = 0, flags = 0, blo = 0;
        AdlerAcc ad{0u, 0u};
        if (have && rp.emit != 0 && (!flat || (sc.comb == 1 && pixp))) word = sc.blk_size[frow + t];
        if (have && !flat) {
            if (rp.emit == 0) {
                const uint64_t b0 = (uint64_t)t * TILE_BM;
                bsz = (uint32_t)min((uint64_t)TILE_BM, sc.nb - b0);
                bdst = (uint32_t)(bitmap_pos + b0);
            } else {
                bsz = word;
                const uint32_t boff = bhdr + sc.blk_off[frow + t];
                bdst = (uint32_t)bitmap_pos + boff;
                if (rp.emit == 8) store_u32_le(rec + bitmap_pos + 16 + 4 * (uint64_t)t, boff);   // blosc bstarts[t]
            }
        }
        if (have && pixp) {
            cnt = sc.tile_cnt[frow + t];
            if (cnt) {
                const ResidGeom q = resid_geom(sc.tile_off[frow + t], cnt, d);
                ps0 = q.ps0;
                if (ADLER) blo = q.b_lo;
                if (sc.comb) {   // (rc_launch.h::residual_src)
                    const uint32_t ro16 = sc.comb == 2 ? (uint32_t)BLK_SLOT / 16 : (word + 15) >> 4, r16 = (cnt * d + 127) >> 7;
                    if (16 * (ro16 + r16) <= sc.blk_stride) rs = 16 * ro16;
                }
                // a tile whose bytes straddle a stored-chunk header of the pixel frame is copied byte by byte behind the loop (bit 31)
                if (q.n && !plain_pos && ((uint64_t)q.b_lo >> ff.chunk_shift) != (((uint64_t)q.b_lo + q.n - 1) >> ff.chunk_shift)) flags = 1u << 31;
                else rn = q.n;
                rdst = (uint32_t)(pix_pos + (plain_pos ? (uint64_t)q.b_lo : stored_pos(ff, q.b_lo)));
            }
        }
        // the first tile behind the item, for the last tile's shared byte (BITS)
        uint32_t ext_cnt = 0, ext_first = 0;
        if (BITS && pixp && t0 + tpi < sc.ntiles) {
            ext_cnt = sc.tile_cnt[frow + t0 + tpi];
            if (ext_cnt)
                ext_first = *reinterpret_cast<const uint32_t *>(residual_src(sc, frow + t0 + tpi, sc.comb == 1 ? sc.blk_size[frow + t0 + tpi] : 0u, ext_cnt, d));
        }
        const uint8_t *slot = rp.emit == 0 ? sc.bitmap + (uint64_t)f * sc.nb_stride + (uint64_t)t0 * TILE_BM : sc.blk_slots + (frow + t0) * sc.blk_stride;
        const uint32_t slot_stride = rp.emit == 0 ? (uint32_t)TILE_BM : sc.blk_stride;
        const uint8_t *rslot = pix_base + (frow + t0) * sc.pix_slot_bytes;

        // ---- pieces -------------------------------------------------------------------------------------------------------------------
        const uint32_t nb16 = (bsz + 15) >> 4, tot = nb16 + ((rn + 15) >> 4);
        const uint32_t incl = wave_incl_scan(tot), cum = incl - tot, T = wave_last(incl);
        const uint32_t packed = bsz | (ps0 << 10) | (rn << 13) | flags;   // bsz <= 644, ps0 <= 7, rn <= 16 384 (15 bits)
        for (uint32_t i0 = 0; 64u * i0 < T; i0 += U) {
            u32x4 x[U];
            uint32_t e[U], dsto[U], meta[U];   // meta: valid bytes | shift << 8
            uint32_t spos[ADLER ? U : 1];      // deflate: the piece's first byte in the residual stream (~0: a piece of a block)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t P = 64u * (i0 + u) + lane;
                uint32_t k = 0;   // the tile of piece P: the last one whose first piece is <= P
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) {
                    const uint32_t c = lane_get(cum, k + step);
                    if (c <= P) k += step;
                }
                const uint32_t ck = lane_get(cum, k), pk = lane_get(packed, k), bd = lane_get(bdst, k), rd = lane_get(rdst, k), rsk = lane_get(rs, k);
                const uint32_t blk = ADLER ? lane_get(blo, k) : 0u;
                const uint32_t bsz_k = pk & 1023u, ps0_k = (pk >> 10) & 7u, rn_k = (pk >> 13) & 0x7FFFu;
                const uint32_t idx = P - ck, nbk = (bsz_k + 15) >> 4;
                const bool isb = idx < nbk;
                const uint32_t j = isb ? idx : idx - nbk;
                const uint32_t left = (isb ? bsz_k : rn_k) - 16u * j;
                const bool act = P < T;
                const uint8_t *src = isb ? slot + (uint64_t)k * slot_stride + 16u * j
                                         : (rsk == 0xFFFFFFFFu ? rslot + (uint64_t)k * sc.pix_slot_bytes : slot + (uint64_t)k * slot_stride + rsk) + 16u * j;
                dsto[u] = (isb ? bd : rd) + 16u * j;
                meta[u] = act ? (min(left, 16u) | ((isb ? 0u : ps0_k) << 8)) : 0u;
                if (ADLER) spos[u] = isb ? 0xFFFFFFFFu : blk + 16u * j;
                x[u] = u32x4{0u, 0u, 0u, 0u};
                e[u] = 0;
                if (act) {
                    x[u] = *reinterpret_cast<const u32x4 *>(src);
                    if (BITS && !isb && ps0_k) e[u] = *reinterpret_cast<const uint32_t *>(src + 16);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (meta[u] & 0xFFu) {
                    const u32x4 y = shift_piece<BITS>(x[u], e[u], meta[u] >> 8);
                    put_piece(rec + dsto[u], y, meta[u] & 0xFFu);
                    if (ADLER && spos[u] != 0xFFFFFFFFu) adler_piece(ad, y, meta[u] & 0xFFu, spos[u]);
                }
        }
        if (pixp) {
        // the residual geometry again (see the bookkeeping)
        ResidGeom q{0u, 0u, 0u, 0u};
        if (cnt) q = resid_geom(sc.tile_off[frow + t], cnt, d);
        // ---- a tile whose residual bytes straddle a stored-chunk header of the pixel frame (once per 4 MiB / 128 KiB): byte by byte --------
        for (uint64_t m = __builtin_amdgcn_ballot_w64((packed >> 31) != 0); m; m &= m - 1) {
            const uint32_t k = (uint32_t)__builtin_ctzll(m);
            const uint32_t n = lane_get(q.n, k), p0 = lane_get(q.ps0, k), b0 = lane_get(q.b_lo, k), rsk = lane_get(rs, k);
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(rsk == 0xFFFFFFFFu ? rslot + (uint64_t)k * sc.pix_slot_bytes : slot + (uint64_t)k * slot_stride + rsk);
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t qb = 8 * i + p0;
                const uint32_t v = __builtin_amdgcn_alignbit(s32[(qb >> 5) + 1], s32[qb >> 5], qb & 31u) & 0xFFu;
                pdst[stored_pos(ff, (uint64_t)b0 + i)] = (uint8_t)v;
                if (ADLER) adler_byte(ad, v, b0 + i);
            }
        }
        if (BITS) {

        // ---- the stream byte a tile shares with its successor(s) --------------------------------------------------------------------------
        const uint32_t fin_avail = q.avail, fin_q = 8 * q.n + q.ps0, fin_b = q.b_lo + q.n;
        const uint8_t *psrc = rs == 0xFFFFFFFFu ? rslot + (uint64_t)lane * sc.pix_slot_bytes : slot + (uint64_t)lane * slot_stride + rs;
        const uint32_t first = cnt ? *reinterpret_cast<const uint32_t *>(psrc) : 0u;
        uint32_t byte = 0, got = 8;
        if (fin_avail) {
            byte = (*reinterpret_cast<const u32_u *>(psrc + (fin_q >> 3)) >> (fin_q & 7u)) & ((1u << fin_avail) - 1u);
            got = fin_avail;
        }
        const uint64_t nonempty = __builtin_amdgcn_ballot_w64(cnt != 0);
        uint32_t cur = lane;   // the last tile of the item whose bits the byte has (64: the first tile behind the item)
        while (__builtin_amdgcn_ballot_w64(got < 8 && cur < 64) != 0) {   // wave-uniform: lane_get needs every lane
            const uint64_t higher = cur >= 63 ? 0ull : nonempty & (~0ull << (cur + 1));
            const uint32_t nxt = higher ? (uint32_t)__builtin_ctzll(higher) : 64u;
            const uint32_t cc = lane_get(cnt, nxt & 63u), fd = lane_get(first, nxt & 63u);
            if (got < 8 && cur < 64) {
                const bool in = nxt < 64;
                const uint32_t c2 = in ? cc : ext_cnt, f2 = in ? fd : ext_first;
                if (in || c2) {
                    const uint32_t take = min(8u - got, c2 * d);
                    byte |= (f2 & ((1u << take) - 1u)) << got;
                    got += take;
                    cur = nxt;
                } else cur = 65;   // the tile behind the item is empty (or the frame ends): tile_next from the item's last tile
            }
        }
        if (got < 8) {   // rare: the chain leaves the item's neighbourhood
            uint32_t tt = cur == 65 ? (t0 + tpi - 1 < sc.ntiles ? sc.tile_next[frow + t0 + tpi - 1] : sc.ntiles) : (t0 + tpi < sc.ntiles ? sc.tile_next[frow + t0 + tpi] : sc.ntiles);
            while (tt < sc.ntiles) {
                const uint32_t cc = sc.tile_cnt[frow + tt];
                const uint32_t fd = *reinterpret_cast<const uint32_t *>(residual_src(sc, frow + tt, sc.comb == 1 ? sc.blk_size[frow + tt] : 0u, cc, d));
                const uint32_t take = min(8u - got, cc * d);
                byte |= (fd & ((1u << take) - 1u)) << got;
                got += take;
                if (got >= 8) break;
                tt = sc.tile_next[frow + tt];
            }
        }
        if (fin_avail) {
            pdst[plain_pos ? (uint64_t)fin_b : stored_pos(ff, fin_b)] = (uint8_t)byte;
            if (ADLER) adler_byte(ad, byte & 0xFFu, fin_b);
        }
        }   // BITS
        }   // pixp
        if (ADLER && !flat) {
            // the item's sums -> the frame's accumulators (zeroed by k_layout); k_zlib_finish, behind this kernel, turns them into the trailers.
            // Relaxed device-scope adds and NO fence: a "last item writes the trailers" form needs a release fence per item, and on this chip
            // a device-scope release writes the XCD's whole L2 back - 4096 of them per batch took 230 us and stretched the reduce kernel
            // running next to them from 0.53 to 0.72 ms (profiles/r06_exp1_trailers.md).
            const uint32_t aux = have ? sc.blk_aux[frow + t] : 0u;
            const uint32_t tA0 = wave_last(wave_incl_scan(aux & 0xFFFFu)), tW0 = wave_last(wave_incl_scan(aux >> 16));
            const uint32_t tA1 = wave_last(wave_incl_scan(ad.A % ADLER_P)), tW1 = wave_last(wave_incl_scan(ad.W));
            if (lane == 0) {
                uint32_t *acc = sc.zl_acc + 8 * (uint64_t)f;
                if (tA0 | tW0) {
                    __hip_atomic_fetch_add(&acc[0], tA0 % ADLER_P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&acc[1], tW0 % ADLER_P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tA1 | tW1) {
                    __hip_atomic_fetch_add(&acc[2], tA1 % ADLER_P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&acc[3], tW1 % ADLER_P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// deflate: the frames' Adler-32 sums -> the two trailers of each record (a thread per frame; the kernel boundary orders it behind k_gather's adds)
__global__ __launch_bounds__(64) void k_zlib_finish(GatherArgs sc, RecordParams rp, uint8_t *__restrict__ out, const uint64_t *__restrict__ rec_off, uint32_t B)
{
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= B || sc.status->code != 0) return;
    const FrameFmt ff = frame_fmt(rp.emit);
    const uint32_t *acc = sc.zl_acc + 8 * (uint64_t)f;
    uint8_t *rec = out + rec_off[f];
    const uint32_t cb = bitmap_hdr(ff, rp.emit, sc.ntiles) + sc.frame_cbytes[f] + ff.end, bitmap_pos = rp.level == 1 ? 16u : 8u;
    store_u32_be(rec + bitmap_pos + cb - 4, adler_from_sums(sc.nb, acc[0], acc[1]));
    if (rp.level == 1 && rp.pix_mode != 2) {
        const uint32_t npk = packed_bytes(sc.frame_nnz[f], rp.depth);
        store_u32_be(rec + bitmap_pos + cb + stored_size(ff, npk) - 4, adler_from_sums(npk, acc[2], acc[3]));
    }
}


void launch_gather(const Scratch &sc, const RecordParams &rp, uint32_t B, uint8_t *out, const uint64_t *rec_off, uint32_t hdr_bitmap, uint32_t hdr_pix,
                   uint32_t batch_seq, hipStream_t s)
{
    // a one-wave workgroup per item of 64 tiles (RC_GATHER_WGS, development builds: fewer workgroups, each walking several items)
    static const char *wgs_env = RC_KNOB("RC_GATHER_WGS");
    // 64 tiles per item; fewer where that leaves the chip without work (configs[0]: nine 512 x 512 frames are nine items of 64 tiles -
    // nine wavefronts walking 64 tiles each took twice k_assemble's time; with eight tiles per item they are 72)
    uint32_t tpi = 64;
    while (tpi > 8 && (uint64_t)B * ((sc.ntiles + tpi - 1) / tpi) < 1024) tpi >>= 1;
    const uint32_t gpf = (sc.ntiles + tpi - 1) / tpi, nitems = gpf * B;
    uint32_t wgs = wgs_env ? (uint32_t)atoi(wgs_env) : 0u;
    if (wgs == 0 || wgs > nitems) wgs = nitems;
    const GatherArgs ga{sc.blk_slots, sc.bitmap, sc.pix_slots, sc.blk_size, sc.blk_off, sc.tile_cnt, sc.tile_off, sc.tile_next, sc.frame_nnz, sc.frame_cbytes,
                        sc.frame_pbytes, sc.blk_aux, sc.zl_acc, sc.pixraw, sc.status, sc.first_err, sc.nb, sc.nb_stride, sc.pixraw_stride, sc.ntiles, sc.blk_stride, sc.pix_slot_bytes, sc.comb};
    const bool bits = rp.level == 1 && rp.depth % 8 != 0;
    if (rp.emit == EMIT_DEFLATE) {   // (the zlib streams' Adler-32 is summed up on the way: the ADLER instantiations)
        if (bits) hipLaunchKernelGGL((k_gather<true, RC_GATHER_UB - 1, true>), dim3(wgs), dim3(64), 0, s, ga, rp, out, rec_off, hdr_bitmap, hdr_pix, batch_seq, gpf, nitems, tpi);
        else hipLaunchKernelGGL((k_gather<false, RC_GATHER_U, true>), dim3(wgs), dim3(64), 0, s, ga, rp, out, rec_off, hdr_bitmap, hdr_pix, batch_seq, gpf, nitems, tpi);
        if (rp.pix_mode != 1) hipLaunchKernelGGL(k_zlib_finish, dim3((B + 63) / 64), dim3(64), 0, s, ga, rp, out, rec_off, B);
    } else if (bits)
        hipLaunchKernelGGL((k_gather<true, RC_GATHER_UB, false>), dim3(wgs), dim3(64), 0, s, ga, rp, out, rec_off, hdr_bitmap, hdr_pix, batch_seq, gpf, nitems, tpi);
    else
        hipLaunchKernelGGL((k_gather<false, RC_GATHER_U, false>), dim3(wgs), dim3(64), 0, s, ga, rp, out, rec_off, hdr_bitmap, hdr_pix, batch_seq, gpf, nitems, tpi);
}

}  // namespace rc
