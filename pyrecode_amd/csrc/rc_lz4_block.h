// rc_lz4_block.h - LZ4 block encoder for one 512-byte block held by one wavefront.
//
// Replaces the reference's `lz4.frame.compress(data, compression_level, store_size=False)` on the packed binary map
// (pyrecode/recode_compressors.py:91, called from recode_writer.py:503-505).  The reference pins no compressed bytes for
// this scheme: the contract is a valid LZ4 frame that any stock decoder expands to the bit-exact input.
// Format: lz4_Block_format.md / lz4_Frame_format.md (v1.6.x).
//
// Encoder (no hash table - the input is a sparse bitmap, > 90 % zero bytes at the target sparsity):
//   every run of >= 5 zero bytes becomes  [literal 0x00][match offset=1, length=run-1]  (an overlapping copy, the format's
//   RLE idiom); everything else is literals.  Two phases, both wave-local:
//   1. position-major, 8 positions per lane: zero-byte mask, run detection by shift/AND arithmetic on a 24-bit window
//      (own 8 bits + 8-bit halos from the neighbour lanes via DPP), giving per lane the bits "match start" and "first
//      literal of a sequence"; their ranks (one packed DPP scan) index two small LDS tables of positions.
//   2. sequence-major, one sequence per lane (a 512-byte block has at most 86): literal run [FL[k], MS[k]) and match
//      [MS[k], FL[k+1]) come from the tables; encoded sizes are prefix-summed with one DPP scan; each lane writes its
//      token, length bytes, literals (copied from the LDS image of the block) and offset into the LDS staging block,
//      which is flushed with coalesced dword stores.
//   Work is therefore proportional to the number of sequences (about 40 per block at 1 % sparsity), not to 512.
//   Block-end rules (last 5 bytes literal, last match starts >= 12 bytes before the end) are met by never matching
//   inside the last 12 bytes.  A block that would not shrink is stored raw.
//
// compression_level >= 1 (the reference hands the level to lz4.frame.compress; the device encoder has two efforts): the EVENT
// parser, lz4_parse_events.  A sparse bitmap block is a chain of units [non-zero byte X][zero run]; almost every X is one of
// the eight single-bit values.  A unit whose X occurred before is copied from the earlier unit with the longest zero run
// (offset = distance, no literals: 3 bytes instead of 5), the match reaches back over the zeros in FRONT of X as far as the
// source has zeros there too, and what is left of a gap is the offset-1 run as before.  Stock liblz4 finds the same repeats
// with its hash table (0.33 of raw on independent 512-byte blocks, 0.25 with an optimal parse); this parse gives 0.29, the
// run-only one 0.375.  One event per lane up to 62 events in the block, two per lane up to 126 (dense maps: 2 % Bernoulli 0.59 -> 0.44 of
// raw, the detector-like clusters 0.62 -> 0.53; stock liblz4 with 64 KiB blocks: 0.43 / 0.47); beyond that the run parser.
#pragma once
#include "rc_device.h"

namespace rc {

#ifdef RC_LZ4_PHASE   // development builds (-DRC_PHASE_TIMING, rc_reduce.hip): s_memtime ticks of the encoder's sub-phases, lane 0 of one workgroup in 64
#define LZ4_PH_BEGIN unsigned long long lzt_ = __builtin_amdgcn_s_memtime();
#define LZ4_PH(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (lane_id() == 0 && (blockIdx.x & 63u) == 0) atomicAdd(&g_phase[8 + (i)], n_ - lzt_); lzt_ = n_; } while (0)
#else
#define LZ4_PH_BEGIN
#define LZ4_PH(i) do { } while (0)
#endif

constexpr int LZ4_BLK = TILE_BM;       // 512
constexpr int LZ4_MAXSEQ = LZ4_BLK / 4 + 2;  // LZ4: runs >= 5 (at most 104 sequences); zstd (rc_zstd_wave.h): runs >= 4 (128)

// wave-private LDS working set of the block encoders (LZ4 here, zstd tokenizer in rc_zstd_wave.h)
struct __attribute__((aligned(16))) Lz4Lds {
    uint8_t raw[LZ4_BLK];              // block image in position order
    uint8_t out[BLK_SLOT];             // staging of what goes to the tile's slot
    uint16_t ms[LZ4_MAXSEQ + 2];       // position of the k-th match start
    uint16_t fl[LZ4_MAXSEQ + 2];       // position of the k-th sequence start (first literal; == ms[k] for a sequence without literals)
    uint16_t off[LZ4_MAXSEQ + 2];      // offset of the k-th match (event parser only; the run parser's offsets are all 1)
};

// bit j (0..3) set iff byte j of x is zero (exact, no borrow artefacts)
__device__ __forceinline__ uint32_t zero_bytes4(uint32_t x)
{
    const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 in every zero byte
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}
__device__ __forceinline__ uint32_t lz4_ext(uint32_t x) { return (x >= 15u ? 1u : 0u) + (x >= 270u ? 1u : 0u); }  // x <= 512
// LZ4 length extension bytes for a field whose value is 15 + r, r <= 497 (lengths never exceed the 512-byte block)
__device__ __forceinline__ uint32_t lz4_emit_len(uint8_t *out, uint32_t o, uint32_t r)
{
    if (r >= 255) { out[o++] = 255; r -= 255; }
    out[o++] = (uint8_t)r;
    return o;
}

// table[k], table[k + 1] <- base + (positions of the at most two set bits of `bits`)
__device__ __forceinline__ void table_put2(uint16_t *table, uint32_t k, uint32_t bits, int base)
{
    if (bits) {
        table[k] = (uint16_t)(base + __builtin_ctz(bits));
        const uint32_t rest = bits & (bits - 1);
        if (rest) table[k + 1] = (uint16_t)(base + __builtin_ctz(rest));
    }
}

// ---- parsers: fill L.ms / L.fl (/ L.off) with the block's matches in position order, return their number nm -----------------
// Sequence k (k < nm) = literals [fl[k], ms[k]) + match [ms[k], fl[k+1]); sequence nm = the closing literals [fl[nm], n).

// Run parser (compression_level 0, dense blocks, the blosc path): every run of >= 5 zero bytes = [literal 00][match offset 1].
__device__ __forceinline__ uint32_t lz4_parse_runs(uint64_t own, uint32_t n, Lz4Lds &L)
{
    const int lane = lane_id();
    const int base = 8 * lane;
    const uint32_t z = zero_bytes4((uint32_t)own) | (zero_bytes4((uint32_t)(own >> 32)) << 4);
    auto below = [&](int lim) -> uint32_t {  // mask of own positions p < lim
        const int rel = lim - base;
        return rel >= 8 ? 0xFFu : (rel <= 0 ? 0u : ((1u << rel) - 1u));
    };
    const uint32_t valid = below((int)n);
    const uint32_t zeff = z & below((int)n - 12);  // never match inside the last 12 bytes
    const uint32_t W = wave_prev(zeff) | (zeff << 8) | (wave_next(zeff) << 16);
    const uint32_t R5 = W & (W >> 1) & (W >> 2) & (W >> 3) & (W >> 4);       // bit i: window positions i..i+4 all zero
    const uint32_t Q = R5 | (R5 << 1) | (R5 << 2) | (R5 << 3) | (R5 << 4);   // inside a zero run of length >= 5
    const uint32_t Mw = Q & (Q << 1);                                        // ... and not the run's first byte
    const uint32_t m = (Mw >> 8) & 0xFFu;                                    // own positions produced by a match
    const uint32_t after = ((m << 1) | ((Mw >> 7) & 1u)) & 0xFFu;            // bit j: position j-1 is a match position
    uint32_t fl = ~m & valid & (after | (lane == 0 ? 1u : 0u));              // first literal of a sequence
    uint32_t ms = m & ~after;                                                // first position of a match
    // ranks of this lane's starts among all starts of the block
    const uint32_t cnt = (uint32_t)__builtin_popcount(ms) | ((uint32_t)__builtin_popcount(fl) << 10);
    const uint32_t inc = wave_incl_scan(cnt);
    const uint32_t tot = wave_last(inc);
    const uint32_t nm = tot & 0x3FFu;  // matches; sequences = nm + 1 (the last one has literals only)
    uint32_t kms = (inc - cnt) & 0x3FFu, kfl = ((inc - cnt) >> 10) & 0x3FFu;
    // a lane's 8 positions hold at most two match starts and two sequence starts (a match is >= 4 bytes long and is
    // preceded by at least one literal): two predicated stores each, no loop
    table_put2(L.ms, kms, ms, base);
    table_put2(L.fl, kfl, fl, base);
    __builtin_amdgcn_wave_barrier();
    return nm;
}

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// per 16-bit half: maximum over all LOWER lanes (0 for lane 0)
__device__ __forceinline__ uint32_t wave_excl_pkmax(uint32_t x)
{
    x = wave_prev(x);
    x = pk_max_u16(x, dpp_zero<0x111>(x));        // row_shr:1
    x = pk_max_u16(x, dpp_zero<0x112>(x));        // row_shr:2
    x = pk_max_u16(x, dpp_zero<0x114>(x));        // row_shr:4
    x = pk_max_u16(x, dpp_zero<0x118>(x));        // row_shr:8
    x = pk_max_u16(x, dpp_zero<0x142, 0xA>(x));   // row_bcast:15 -> rows 1 and 3
    x = pk_max_u16(x, dpp_zero<0x143, 0xC>(x));   // row_bcast:31 -> rows 2 and 3
    return x;
}

constexpr int LZ4_EV_MAX = 62;   // events per block the event parser takes with one event per lane (lane 0 is the block start: at most 2 * 63 matches)
constexpr int LZ4_EV_MAX2 = 126; // ... with two events per lane (lz4_parse_events2: dense maps - 2 % Bernoulli, the detector-like clusters)
constexpr int LZ4_NM_MAX2 = 126; // ... and the matches it may leave (the position tables and phase 2's two rounds; dense blocks have 40 - 60)
// A block with more non-zero bytes than this is stored without a parse (compression_level >= 1): beyond ~9 % of the pixels set a 512-byte
// block of a binary map no longer shrinks under LZ4 (Bernoulli maps, tests/lz4_parse_model.py: at 10 % the stream is 0.9955 of raw and half the
// blocks do not shrink at all, at 12 % 0.9993), while the run parser and phase 2 still cost their ~180 vector instructions per tile and
// frame.  The threshold gives up < 0.5 % of the stream at 9 - 10 % and nothing measurable elsewhere.
constexpr int LZ4_NZ_STORE = 272;

// Event parser (compression_level >= 1).  Lane 0 stands for the block start ("event" at position -1), lane k >= 1 for the
// k-th non-zero byte X_k at p_k, followed by R_k zeros.  Per lane, with j = the earlier event of the same single-bit value
// that has the longest zero run (one packed prefix-max over the lanes, keys = min(R, 127) << 6 | lane, a 16-bit field per value):
//   trail = min(R_k, R_j)                              zeros behind X_k the source also has        (match only if >= 3)
//   lead  = min(R_{k-1} - trail_{k-1}, R_{j-1})        zeros in front of X_k, not covered by event k-1's match, that the
//                                                      source has in front of X_j as well
//   match 1 = [p_k - lead, p_k + 1 + trail)  at offset p_k - p_j, no literals
//   match 2 = what is left of the gap behind it, [.., p_{k+1} - lead_{k+1}), at offset 1 - or, when event k has no match 1 (an
//             offset-1 copy then needs one LITERAL zero in front of it): the whole gap copied from inside the longest zero run of
//             the earlier lanes (one more prefix-max, key = min(R, 511) << 6 | lane) when that run is long enough - no literal;
//             liblz4's hash chains find the same source (round 4: bitmap stream 0.293 -> 0.283 of raw at 1 %)
// Returns nm, or 0xFFFFFFFF when the block holds more than LZ4_EV_MAX2 events (the caller runs lz4_parse_runs); 63 .. 126 events:
// lz4_parse_events2 below.
__device__ __forceinline__ uint32_t lz4_parse_events2(uint32_t nev, uint32_t n, Lz4Lds &L);
__device__ __forceinline__ uint32_t lz4_parse_events(uint64_t own, uint32_t n, Lz4Lds &L, uint32_t &nev_out)
{
    const int lane = lane_id();
    const int base = 8 * lane;
    uint16_t *ev = reinterpret_cast<uint16_t *>(L.out);   // [k] = p_k + 1 (k = 0: 0), [nev + 1] = n + 1; dead before phase 2 writes L.out
    const uint32_t z = zero_bytes4((uint32_t)own) | (zero_bytes4((uint32_t)(own >> 32)) << 4);
    const int rel = (int)n - base;
    const uint32_t valid = rel >= 8 ? 0xFFu : (rel <= 0 ? 0u : ((1u << rel) - 1u));
    uint32_t nz = ~z & valid;
    const uint32_t cnt = (uint32_t)__builtin_popcount(nz);
    const uint32_t inc = wave_incl_scan(cnt);
    const uint32_t nev = wave_last(inc);
    nev_out = nev;
    if (nev > (uint32_t)LZ4_EV_MAX2) return 0xFFFFFFFFu;
    for (uint32_t k = inc - cnt + 1; nz; nz &= nz - 1) ev[k++] = (uint16_t)(base + __builtin_ctz(nz) + 1);
    if (lane == 0) { ev[0] = 0; ev[nev + 1] = (uint16_t)(n + 1); }
    __builtin_amdgcn_wave_barrier();
    if (nev > (uint32_t)LZ4_EV_MAX) return lz4_parse_events2(nev, n, L);
    const bool act = (uint32_t)lane <= nev;
    const uint32_t e0 = ev[lane], e1 = ev[lane + 1];
    const uint32_t P1 = act ? e0 : n + 1;                 // p + 1: the first byte behind X
    const uint32_t Pn = act ? e1 : n + 1;                 // p_next + 1
    const uint32_t X = L.raw[P1 ? P1 - 1 : 0];
    const bool classed = act && lane > 0 && (X & (X - 1)) == 0;   // (X != 0 by construction)
    const uint32_t cls = (uint32_t)__builtin_ctz(X | 0x100u);
    const uint32_t Rk = Pn - P1 - 1 + (act ? 0u : 1u);    // zeros behind X (inactive lanes: 0)
    // the best earlier source per value: exclusive prefix maximum of the keys, four registers of two 16-bit fields
    const uint32_t key = classed ? ((min(Rk, 127u) << 6) | (uint32_t)lane) << (16 * (cls & 1u)) : 0u;
    const uint32_t a0 = wave_excl_pkmax((cls >> 1) == 0 ? key : 0u), a1 = wave_excl_pkmax((cls >> 1) == 1 ? key : 0u);
    const uint32_t a2 = wave_excl_pkmax((cls >> 1) == 2 ? key : 0u), a3 = wave_excl_pkmax((cls >> 1) == 3 ? key : 0u);
    const uint32_t pair = (cls >> 1) == 0 ? a0 : ((cls >> 1) == 1 ? a1 : ((cls >> 1) == 2 ? a2 : a3));
    const uint32_t best = classed ? ((pair >> (16 * (cls & 1u))) & 0xFFFFu) : 0u;
    const uint32_t j = best & 63u;
    // from lane j: its run, the run in front of it, its position
    const uint32_t w = Rk | (wave_prev(Rk) << 10) | (P1 << 20);
    const uint32_t wj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)w);
    const int Rj = (int)(wj & 0x3FFu), Rjm1 = (int)((wj >> 10) & 0x3FFu);
    const uint32_t P1j = wj >> 20;
    int trail = min(min((int)Rk, Rj), (int)n - 5 - (int)P1);                  // a match ends at least 5 bytes before the block's end
    const bool has = best != 0 && P1 + 11 <= n && trail >= 3;                // ... and starts at least 12 bytes before it
    trail = has ? trail : 0;
    const int d = (int)Rk - trail;
    const int dprev = (int)wave_prev((uint32_t)d);        // (cross-lane reads stay outside the selects: every lane takes part)
    const int lead = has ? min(dprev, Rjm1) : 0;
    const int leadn = (int)wave_next((uint32_t)lead);
    int gs = (int)P1 + trail + (has ? 0 : 1);
    const int ge = min((int)Pn - 1 - leadn, (int)n - 5);
    // the longest zero run of the lanes in front (lane 0's is the block's leading zeros), and where it starts
    const uint32_t zbest = wave_excl_pkmax(act ? ((min(Rk, 511u) << 6) | (uint32_t)lane) : 0u) & 0xFFFFu;
    const uint32_t P1z = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((zbest & 63u) << 2), (int)w) >> 20;
    const int L0 = ge - (int)P1;
    const bool zsrc = act && lane > 0 && !has && L0 >= 4 && (int)(zbest >> 6) >= L0 && (int)P1 + 12 <= (int)n;
    uint32_t off2 = 1;
    if (zsrc) { gs = (int)P1; off2 = P1 - P1z; }
    const bool rle = act && gs + 12 <= (int)n && ge - gs >= 4;
    const uint32_t c = (has ? 1u : 0u) + (rle ? 1u : 0u);
    const uint32_t rinc = wave_incl_scan(c);
    const uint32_t nm = wave_last(rinc);
    uint32_t r = rinc - c;
    if (lane == 0) L.fl[0] = 0;
    if (has) {
        L.ms[r] = (uint16_t)((int)P1 - 1 - lead);
        L.fl[r + 1] = (uint16_t)((int)P1 + trail);
        L.off[r] = (uint16_t)(P1 - P1j);
        ++r;
    }
    if (rle) {
        L.ms[r] = (uint16_t)gs;
        L.fl[r + 1] = (uint16_t)ge;
        L.off[r] = (uint16_t)off2;
    }
    __builtin_amdgcn_wave_barrier();
    return nm;
}

// The event parser with TWO events per lane (63 .. 126 events in the block): lane l stands for events 2l and 2l + 1 (event 0 = the block
// start), the rules are lz4_parse_events' own.  The prefix maxima run over the lanes' JOINED keys (the same five scans), a lane's second
// event also looks at its first; event numbers take 7 bits of a key (min(R, 127) << 7 | k, min(R, 511) << 7 | k); what an event needs of
// its source (run, run in front, position) comes from a 128-entry LDS table that borrows the position tables' space until they are written.
// ev[] = the list lz4_parse_events has filled.  Returns nm, or 0xFFFFFFFF (more than LZ4_NM_MAX2 matches: the run parser).
__device__ __forceinline__ uint32_t lz4_parse_events2(uint32_t nev, uint32_t n, Lz4Lds &L)
{
    static_assert(sizeof(L.ms) + sizeof(L.fl) >= 128 * 4 && offsetof(Lz4Lds, fl) == offsetof(Lz4Lds, ms) + sizeof(L.ms) && offsetof(Lz4Lds, ms) % 4 == 0, "source table");
    const int lane = lane_id();
    const uint16_t *ev = reinterpret_cast<const uint16_t *>(L.out);
    uint32_t *W = reinterpret_cast<uint32_t *>(L.ms);
    const uint32_t ka = 2u * (uint32_t)lane, kb = ka + 1u;
    const bool acta = ka <= nev, actb = kb <= nev;
    const uint32_t e0 = ev[ka], e1 = ev[ka + 1], e2 = ev[ka + 2];
    const uint32_t P1a = acta ? e0 : n + 1, P1b = actb ? e1 : n + 1;       // p + 1: the first byte behind X
    const uint32_t Pna = P1b, Pnb = (kb + 1 <= nev + 1 && actb) ? e2 : n + 1;
    const uint32_t Xa = L.raw[P1a ? P1a - 1 : 0], Xb = L.raw[P1b - 1 < n ? P1b - 1 : 0];
    const bool cla = acta && lane > 0 && (Xa & (Xa - 1)) == 0, clb = actb && (Xb & (Xb - 1)) == 0;
    const uint32_t ca = (uint32_t)__builtin_ctz(Xa | 0x100u), cb = (uint32_t)__builtin_ctz(Xb | 0x100u);
    const uint32_t Ra = Pna - P1a - 1 + (acta ? 0u : 1u), Rb = Pnb - P1b - 1 + (actb ? 0u : 1u);   // zeros behind X (inactive: 0)
    const uint32_t keya = cla ? ((min(Ra, 127u) << 7) | ka) : 0u, keyb = clb ? ((min(Rb, 127u) << 7) | kb) : 0u;
    const uint32_t fa = keya << (16 * (ca & 1u)), fb = keyb << (16 * (cb & 1u));
    uint32_t A[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) A[q] = wave_excl_pkmax(pk_max_u16((ca >> 1) == q ? fa : 0u, (cb >> 1) == q ? fb : 0u));
    const uint32_t paira = (ca >> 1) == 0 ? A[0] : ((ca >> 1) == 1 ? A[1] : ((ca >> 1) == 2 ? A[2] : A[3]));
    const uint32_t pairb = (cb >> 1) == 0 ? A[0] : ((cb >> 1) == 1 ? A[1] : ((cb >> 1) == 2 ? A[2] : A[3]));
    const uint32_t besta = cla ? ((paira >> (16 * (ca & 1u))) & 0xFFFFu) : 0u;
    const uint32_t bestb = clb ? max((pairb >> (16 * (cb & 1u))) & 0xFFFFu, (cla && ca == cb) ? keya : 0u) : 0u;
    // every event's run, the run in front of it and its position, by event number
    const uint32_t Rbprev = wave_prev(Rb);
    W[ka] = Ra | (Rbprev << 10) | (P1a << 20);
    W[kb] = Rb | (Ra << 10) | (P1b << 20);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const uint32_t wja = W[besta & 127u], wjb = W[bestb & 127u];
    const uint32_t za = acta ? ((min(Ra, 511u) << 7) | ka) : 0u, zb = actb ? ((min(Rb, 511u) << 7) | kb) : 0u;
    const uint32_t Z = wave_excl_pkmax(max(za, zb)) & 0xFFFFu;             // the longest zero run in front of the lane (and where it starts)
    const uint32_t zbesta = Z, zbestb = max(Z, za);
    const uint32_t P1za = W[zbesta & 127u] >> 20, P1zb = W[zbestb & 127u] >> 20;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();                                          // (the table's space becomes the position tables' again)
    int traila = min(min((int)Ra, (int)(wja & 0x3FFu)), (int)n - 5 - (int)P1a), trailb = min(min((int)Rb, (int)(wjb & 0x3FFu)), (int)n - 5 - (int)P1b);
    const bool hasa = besta != 0 && P1a + 11 <= n && traila >= 3, hasb = bestb != 0 && P1b + 11 <= n && trailb >= 3;
    traila = hasa ? traila : 0;
    trailb = hasb ? trailb : 0;
    const int da = (int)Ra - traila, db = (int)Rb - trailb;
    const int dpreva = (int)wave_prev((uint32_t)db);
    const int leada = hasa ? min(dpreva, (int)((wja >> 10) & 0x3FFu)) : 0, leadb = hasb ? min(da, (int)((wjb >> 10) & 0x3FFu)) : 0;
    const int leadna = leadb, leadnb = (int)wave_next((uint32_t)leada);
    int gsa = (int)P1a + traila + (hasa ? 0 : 1), gsb = (int)P1b + trailb + (hasb ? 0 : 1);
    const int gea = min((int)Pna - 1 - leadna, (int)n - 5), geb = min((int)Pnb - 1 - leadnb, (int)n - 5);
    const int L0a = gea - (int)P1a, L0b = geb - (int)P1b;
    const bool zsa = acta && lane > 0 && !hasa && L0a >= 4 && (int)(zbesta >> 7) >= L0a && (int)P1a + 12 <= (int)n;
    const bool zsb = actb && !hasb && L0b >= 4 && (int)(zbestb >> 7) >= L0b && (int)P1b + 12 <= (int)n;
    uint32_t off2a = 1, off2b = 1;
    if (zsa) { gsa = (int)P1a; off2a = P1a - P1za; }
    if (zsb) { gsb = (int)P1b; off2b = P1b - P1zb; }
    const bool rlea = acta && gsa + 12 <= (int)n && gea - gsa >= 4, rleb = actb && gsb + 12 <= (int)n && geb - gsb >= 4;
    const uint32_t c = (hasa ? 1u : 0u) + (rlea ? 1u : 0u) + (hasb ? 1u : 0u) + (rleb ? 1u : 0u);
    const uint32_t rinc = wave_incl_scan(c);
    const uint32_t nm = wave_last(rinc);
    if (nm > (uint32_t)LZ4_NM_MAX2) return 0xFFFFFFFFu;
    uint32_t r = rinc - c;
    if (lane == 0) L.fl[0] = 0;
    if (hasa) {
        L.ms[r] = (uint16_t)((int)P1a - 1 - leada);
        L.fl[r + 1] = (uint16_t)((int)P1a + traila);
        L.off[r] = (uint16_t)(P1a - (wja >> 20));
        ++r;
    }
    if (rlea) {
        L.ms[r] = (uint16_t)gsa;
        L.fl[r + 1] = (uint16_t)gea;
        L.off[r] = (uint16_t)off2a;
        ++r;
    }
    if (hasb) {
        L.ms[r] = (uint16_t)((int)P1b - 1 - leadb);
        L.fl[r + 1] = (uint16_t)((int)P1b + trailb);
        L.off[r] = (uint16_t)(P1b - (wjb >> 20));
        ++r;
    }
    if (rleb) {
        L.ms[r] = (uint16_t)gsb;
        L.fl[r + 1] = (uint16_t)geb;
        L.off[r] = (uint16_t)off2b;
    }
    __builtin_amdgcn_wave_barrier();
    return nm;
}

// Wave-collective.  Precondition: L.raw holds the block in position order (written by this wavefront) and `own` is this
// lane's 8 bytes raw[8*lane .. 8*lane+8), little-endian.  n: valid bytes (1..512).  EVENTS: the event parser (else runs only).
// Returns the compressed size (wave-uniform); the payload is in L.out[4 .. 4 + size) only when size < n (the four bytes in
// front of it take the block's size word: lz4_stage_slot completes the image of the tile's slot in place).
template <bool EVENTS = false>
__device__ __forceinline__ uint32_t lz4_encode_block(uint64_t own, uint32_t n, Lz4Lds &L)
{
    uint8_t *const pay = L.out + 4;
    const int lane = lane_id();
    // ---- phase 1: the parse ----------------------------------------------------------------------------------------------
    uint32_t nm = 0xFFFFFFFFu;
    bool off1 = true;
    LZ4_PH_BEGIN
    if (EVENTS) {
        uint32_t nev = 0;
        nm = lz4_parse_events(own, n, L, nev);
        off1 = nm == 0xFFFFFFFFu;
        if (nev > (uint32_t)LZ4_NZ_STORE) return n;   // hopeless (see LZ4_NZ_STORE): the caller stores the block
    }
    LZ4_PH(0);
    if (nm == 0xFFFFFFFFu) nm = lz4_parse_runs(own, n, L);
    LZ4_PH(1);

    // ---- phase 2: one sequence per lane ------------------------------------------------------------------------------
    uint32_t carry = 0, total = 0;
    uint32_t seq_o[2], seq_fs[2], seq_ll[2], seq_ml4[2], seq_off[2];  // at most 2 rounds of 64 sequences (runs: 86, events: 127)
    const uint32_t nrounds = (nm + 64) / 64;              // ceil((nm + 1) / 64)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        seq_ll[rd] = 0xFFFFFFFFu;  // "no sequence"
        if ((uint32_t)rd < nrounds) {
            const uint32_t k = rd * 64 + lane;
            uint32_t size = 0;
            if (k <= nm) {
                // the table reads are issued together (one LDS round trip); entries behind the tables' ends are never used
                const uint32_t fs = L.fl[k], msk = L.ms[k], fnext = L.fl[k + 1];
                seq_off[rd] = EVENTS && !off1 ? (uint32_t)L.off[k] : 1u;
                const uint32_t q = k < nm ? msk : n;
                const uint32_t ll = q - fs;
                size = 1 + lz4_ext(ll) + ll;
                uint32_t ml4 = 0xFFFFu;  // final sequence: no match
                if (k < nm) {
                    ml4 = fnext - q - 4u;
                    size += 2 + lz4_ext(ml4);
                }
                seq_fs[rd] = fs; seq_ll[rd] = ll; seq_ml4[rd] = ml4;
            }
            const uint32_t sinc = wave_incl_scan(size);
            seq_o[rd] = carry + sinc - size;
            carry += wave_last(sinc);
        }
    }
    total = carry;
    LZ4_PH(2);
    if (total >= n) return total;  // would not shrink: caller stores the block raw
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        if ((uint32_t)rd < nrounds && seq_ll[rd] != 0xFFFFFFFFu) {
            uint32_t o = seq_o[rd];
            const uint32_t ll = seq_ll[rd], ml4 = seq_ml4[rd], fs = seq_fs[rd];
            pay[o++] = (uint8_t)((min(ll, 15u) << 4) | (ml4 == 0xFFFFu ? 0u : min(ml4, 15u)));
            if (ll >= 15) o = lz4_emit_len(pay, o, ll - 15);
            // literals: 4 bytes per step from two aligned dwords of the block image (one LDS round trip per step; a
            // byte-by-byte copy pays one per byte, and the longest literal run of the block sets the trip count)
            {
                const uint32_t *raw32 = reinterpret_cast<const uint32_t *>(L.raw);
                for (uint32_t i = 0; i < ll; i += 4) {
                    const uint32_t a = (fs + i) >> 2;
                    const uint32_t v = __builtin_amdgcn_alignbyte(raw32[a + 1], raw32[a], (fs + i) & 3u);  // (a + 1 may be pay[0..3]: unused bytes)
                    const uint32_t rem = ll - i;
                    pay[o + i] = (uint8_t)v;
                    if (rem > 1) pay[o + i + 1] = (uint8_t)(v >> 8);
                    if (rem > 2) pay[o + i + 2] = (uint8_t)(v >> 16);
                    if (rem > 3) pay[o + i + 3] = (uint8_t)(v >> 24);
                }
            }
            o += ll;
            if (ml4 != 0xFFFFu) {
                pay[o++] = (uint8_t)seq_off[rd];  // offset, little-endian
                pay[o++] = (uint8_t)(seq_off[rd] >> 8);
                if (ml4 >= 15) o = lz4_emit_len(pay, o, ml4 - 15);
            }
        }
    }
    LZ4_PH(3);   // (measured and dropped in round 5: literal runs of more than four bytes copied by the whole wave, a byte per lane - nothing on the
                 //  sparse headline, 4.6 % SLOWER on the detector-like stack whose blocks hold many such runs: profiles/r05_exp9_level2_resting_nodes_ab.log;
                 //  in round 4: a sequence's bytes put together in a register and written with byte-aligned ds_write_b32 /
                 //  b16 - the compiler emits them, LDS runs in unaligned access mode - instead of one byte-write each: 1-2 % SLOWER everywhere)
    return total;
}

// blosc1 bit-shuffle of one block with typesize 8 (bitshuffle's bshuf_trans_bit_elem, little-endian bit order): with S
// elements, output row r (r = 0..63: bit r%8 of byte r/8 of every element) is S/8 bytes whose bit i is that bit of element
// i; S is the element count rounded down to a multiple of 8, the bytes behind the shuffled part stay as they are
// (c-blosc's blosc_internal_bitshuffle).  A lane owns one element (its 8 consecutive bytes), so row r is the wave ballot
// of bit r.  Wave-collective; precondition: L.raw holds the plain block and `elem` is this lane's 8 bytes of it.
// Leaves the shuffled block in L.raw and returns this lane's 8 bytes of it.
__device__ __forceinline__ uint64_t bitshuffle_block(uint64_t elem, uint32_t n, Lz4Lds &L)
{
    const int lane = lane_id();
    const uint32_t S = (n >> 3) & ~7u;        // elements taking part in the bit transpose
    if (S == 0) return elem;
    uint32_t *raw32 = reinterpret_cast<uint32_t *>(L.raw);
    const uint32_t pc = (uint32_t)__builtin_popcountll(elem);
    const bool sparse = S == 64 && __builtin_amdgcn_ballot_w64(pc > 8) == 0;
    if (sparse) {
        // the transposed matrix is as sparse as the block: clear it, then every lane ORs its few set bits into place
        // (bit r of element i -> row r = bytes [8r, 8r+8), bit i of the row)
        raw32[2 * lane] = 0; raw32[2 * lane + 1] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint64_t q = elem; q; q &= q - 1) {
            const uint32_t r = (uint32_t)__builtin_ctzll(q);
            __hip_atomic_fetch_or(&raw32[2 * r + ((uint32_t)lane >> 5)], 1u << (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    } else {
        const uint32_t rowb = S >> 3;         // bytes per output row
        const uint64_t in_s = (uint32_t)lane < S ? elem : 0ull;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const uint64_t row = __builtin_amdgcn_ballot_w64(((in_s >> r) & 1ull) != 0);
            if ((uint32_t)lane < rowb) L.raw[r * rowb + lane] = (uint8_t)(row >> (8 * lane));
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t lo = raw32[2 * lane], hi = raw32[2 * lane + 1];
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// Wave-collective: complete the image of the tile's slot in L.out = [u32 block word][payload]: the word in front of a compressed
// payload (already at L.out + 4), or word + the raw bytes for a block that did not shrink.  LZ4 frame: bit 31 of the word marks
// a stored block; blosc: a stored block is marked by word == n.  Returns the image's bytes (4 + payload).
__device__ __forceinline__ uint32_t lz4_stage_slot(uint64_t own, uint32_t n, uint32_t csize, Lz4Lds &L, bool blosc = false)
{
    uint32_t *out32 = reinterpret_cast<uint32_t *>(L.out);
    const int lane = lane_id();
    if (csize >= n) {
        if (lane == 0) out32[0] = blosc ? n : (n | 0x80000000u);
        if ((uint32_t)(8 * lane) < n) { out32[1 + 2 * lane] = (uint32_t)own; out32[2 + 2 * lane] = (uint32_t)(own >> 32); }
        __builtin_amdgcn_wave_barrier();
        return 4 + n;
    }
    if (lane == 0) out32[0] = csize;
    __builtin_amdgcn_wave_barrier();
    return 4 + csize;
}

// Wave-collective: stage + write the slot as whole 128-byte lines (slots are 128-byte aligned, BLK_SLOT is a multiple of 128;
// the bytes behind the image are unused slot space, and partial-line writes cost a read-modify-write at the memory side).
// Returns slot bytes used.  (The fused reduce kernel stages here and writes through its own combined store, rc_reduce.hip.)
__device__ __forceinline__ uint32_t lz4_store_block(uint8_t *slot, uint64_t own, uint32_t n, uint32_t csize, Lz4Lds &L, bool blosc = false)
{
    const uint32_t used = lz4_stage_slot(own, n, csize, L, blosc);
    const uint32_t *p = reinterpret_cast<const uint32_t *>(L.out);
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(slot);
    const uint32_t ndw = min((((used + 3) >> 2) + 31u) & ~31u, (uint32_t)BLK_SLOT / 4);
    for (uint32_t i = lane_id(); i < ndw; i += 64) slot32[i] = p[i];
    return used;
}

}  // namespace rc
