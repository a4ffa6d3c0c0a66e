// rc_deflate_block.h - DEFLATE block encoder for one 512-byte block held by one wavefront (compression_scheme 0 on the device).
//
// Replaces the reference's `zlib.compress(data, compression_level)` on the packed binary map (pyrecode/recode_compressors.py:84-85,
// called from recode_writer.py:503-505) for a ctx created with RC_SCHEME_ZLIB_DEVICE.  The host path (the same stdlib call the
// reference makes, byte-identical files: G3 / G9) stays the default; what this encoder promises is what every other device codec
// promises: a valid zlib stream (RFC 1950 / 1951) that any stock inflate expands to the bit-exact input.
//
// A tile's 512 map bytes become ONE fixed-Huffman block (BTYPE 01), closed by an empty stored block - zlib's own Z_SYNC_FLUSH
// marker: 3 header bits, padding to the byte, 00 00 FF FF - so that every tile's image is a whole number of bytes and the frames'
// streams are put together by the byte-granular k_gather like every other codec's; the frame's last tile carries BFINAL and is
// padded to the byte instead.  A tile that would not shrink is a stored block (5 + n bytes).
//   parse:  rc_lz4_block.h's event / run parsers - a sparse map is a chain of units [non-zero byte][zero run]; a unit whose byte
//           occurred before is ONE match (length / distance pair) from the earlier unit with the longest run, what is left of a gap
//           a distance-1 run (or a copy from inside the longest earlier zero run).  LZ4's block-end rules cost DEFLATE nothing it
//           needs; matches of more than 258 bytes are written as two.
//   emit:   sequence-major, one sequence (literals + match) per lane: bit sizes -> one wave scan -> every lane shifts its codes into
//           a 64-bit accumulator and ORs whole dwords into the zeroed LDS image (LDS atomics; neighbouring lanes share a dword).
// Serial restatement judged by stdlib zlib: tests/deflate_block_model.py, tests/test_deflate_format_cpu.py; the device's bytes equal
// the model's tile for tile (tests/test_gpu_parity.py).
// The stream's Adler-32 (of the UNcompressed map) comes from per-tile partials left here: see deflate_adler_word.
#pragma once
#include "rc_lz4_block.h"

namespace rc {

constexpr uint32_t ADLER_P = 65521u;

// ---- bit sink: codes of up to 32 bits into a zeroed LDS image ------------------------------------------------------------------
struct BitSink {
    uint32_t *out32;
    uint32_t w, nb;     // dword index, bits of `acc` in use (< 32 between calls)
    uint64_t acc;
};
__device__ __forceinline__ BitSink sink_at(uint32_t *out32, uint32_t bitpos) { return BitSink{out32, bitpos >> 5, bitpos & 31u, 0ull}; }
__device__ __forceinline__ void sink_put(BitSink &s, uint32_t code, uint32_t len)   // code < 2^len, len <= 32
{
    s.acc |= (uint64_t)code << s.nb;
    s.nb += len;
    if (s.nb >= 32) {
        __hip_atomic_fetch_or(&s.out32[s.w], (uint32_t)s.acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        ++s.w;
        s.acc >>= 32;
        s.nb -= 32;
    }
}
__device__ __forceinline__ void sink_flush(BitSink &s)
{
    if (s.nb) __hip_atomic_fetch_or(&s.out32[s.w], (uint32_t)s.acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// ---- the fixed code (RFC 1951 3.2.6); Huffman codes are packed starting with their most significant bit ----------------------------
__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t len) { return __builtin_bitreverse32(v) >> (32u - len); }
// literal b: 8 bits (00110000 + b) below 144, 9 bits (110010000 + b - 144) from there
__device__ __forceinline__ uint32_t deflate_lit(uint32_t b, uint32_t &len)
{
    const bool hi = b >= 144u;
    len = hi ? 9u : 8u;
    return rev_bits(hi ? 0x190u + b - 144u : 0x30u + b, len);
}
// one match (3 <= length <= 258, 1 <= dist <= 512): length symbol + extra bits, distance symbol + extra bits, as one code of <= 25 bits
__device__ __forceinline__ uint32_t deflate_match(uint32_t length, uint32_t dist, uint32_t &len)
{
    const uint32_t l = length - 3u;
    uint32_t e = 0, sym = 257u + l;
    if (l >= 8u) {
        e = (31u - (uint32_t)__builtin_clz(l)) - 2u;
        sym = 257u + 4u * (e + 1u) + ((l >> e) & 3u);
        if (l == 255u) { e = 0; sym = 285u; }
    }
    const uint32_t sl = sym < 280u ? 7u : 8u;
    uint32_t code = rev_bits(sym < 280u ? sym - 256u : 0xC0u + sym - 280u, sl) | ((l & ((1u << e) - 1u)) << sl);
    uint32_t n = sl + e;
    const uint32_t d = dist - 1u;
    uint32_t de = 0, ds = d;
    if (d >= 4u) {
        const uint32_t hb = 31u - (uint32_t)__builtin_clz(d);
        de = hb - 1u;
        ds = 2u * hb + ((d >> de) & 1u);
    }
    code |= (rev_bits(ds, 5u) | ((d & ((1u << de) - 1u)) << 5)) << n;
    len = n + 5u + de;
    return code;
}
// a parser's match of up to 512 bytes as one or two DEFLATE matches: the first one's length (the second is what is left, >= 3)
__device__ __forceinline__ uint32_t deflate_first_part(uint32_t length)
{
    return length <= 258u ? length : (length - 258u >= 3u ? 258u : length - 3u);
}

// bytes >= 144 among the `ll` bytes from raw[fs] on (their literals take 9 bits)
__device__ __forceinline__ uint32_t deflate_count_hi(const uint8_t *raw, uint32_t fs, uint32_t ll)
{
    const uint32_t *raw32 = reinterpret_cast<const uint32_t *>(raw);
    uint32_t c = 0;
    for (uint32_t i = 0; i < ll; i += 4) {
        const uint32_t a = (fs + i) >> 2;
        const uint32_t v = __builtin_amdgcn_alignbyte(raw32[a + 1], raw32[a], (fs + i) & 3u);   // (a + 1 may lie behind raw: masked below)
        uint32_t hi = v & ((v & 0x70707070u) + 0x70707070u) & 0x80808080u;                       // bit 7 and one of bits 4..6: >= 0x90
        const uint32_t rem = ll - i;
        if (rem < 4) hi &= (1u << (8 * rem)) - 1u;
        c += (uint32_t)__builtin_popcount(hi);
    }
    return c;
}

// Adler-32 of the frame's map from per-tile partials (RFC 1950): with A = sum of the bytes and W = sum of i * b_i over the stream
// positions i,  s1 = 1 + A,  s2 = n + n A - W  (mod 65521).  A tile leaves  (A_t mod p) | ((512 t A_t + W_t) mod p) << 16  where
// W_t weighs its bytes by their position inside the tile: the frame's sums are then plain sums of the tiles' halves
// (k_gather adds them up, rc_gather.hip).  Wave-collective; `own` = the lane's 8 bytes (bytes behind the map's end are zero).
__device__ __forceinline__ uint32_t deflate_adler_word(uint64_t own, uint32_t tile)
{
    const uint32_t lo = (uint32_t)own, hi = (uint32_t)(own >> 32), lane = (uint32_t)lane_id();
    const uint32_t a = __builtin_amdgcn_sad_u8(hi, 0u, __builtin_amdgcn_sad_u8(lo, 0u, 0u));
    const uint32_t q = __builtin_amdgcn_udot4(hi, 0x07060504u, __builtin_amdgcn_udot4(lo, 0x03020100u, 0u, false), false);
    const uint32_t A = wave_last(wave_incl_scan(a));                          // <= 512 * 255
    const uint32_t W = wave_last(wave_incl_scan(8u * lane * a + q));          // <= 255 * 511 * 512 / 2 < 2^25
    const uint32_t Am = A % ADLER_P;
    const uint32_t Wm = ((((512u * tile) % ADLER_P) * Am) % ADLER_P + W) % ADLER_P;
    return Am | (Wm << 16);
}

// the trailer of a zlib stream of n bytes whose byte sum is A and position-weighted byte sum W (any representatives mod 65521); it is
// stored big-endian (RFC 1950)
__device__ __forceinline__ uint32_t adler_from_sums(uint64_t n, uint32_t A, uint32_t W)
{
    const uint64_t a = A % ADLER_P, w = W % ADLER_P, nm = n % ADLER_P;
    const uint32_t s1 = (uint32_t)((1 + a) % ADLER_P);
    const uint32_t s2 = (uint32_t)((nm * (1 + a) + ADLER_P - w) % ADLER_P);
    return (s2 << 16) | s1;
}
__device__ __forceinline__ void store_u32_be(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

// OR v (a bit string of up to 64 bits, zero above its end) into the zeroed image at bit `pos`: three dwords, unconditionally (an OR of
// zero changes nothing, and three LDS instructions are cheaper than the exec-mask branches that would skip them; pos + 96 must lie
// inside the image's allocation).  Neighbouring lanes share dwords: LDS atomics.
__device__ __forceinline__ void or_bits64(uint32_t *out32, uint32_t pos, uint32_t lo, uint32_t hi)
{
    const uint32_t w = pos >> 5, sh = pos & 31u;
    const uint64_t a = (uint64_t)lo << sh, b = (uint64_t)hi << sh;
    __hip_atomic_fetch_or(&out32[w], (uint32_t)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_or(&out32[w + 1], (uint32_t)(a >> 32) | (uint32_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_or(&out32[w + 2], (uint32_t)(b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void or_bits32(uint32_t *out32, uint32_t pos, uint32_t v)
{
    const uint32_t w = pos >> 5, sh = pos & 31u;
    const uint64_t a = (uint64_t)v << sh;
    __hip_atomic_fetch_or(&out32[w], (uint32_t)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_or(&out32[w + 1], (uint32_t)(a >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Wave-collective.  Precondition: L.raw holds the block in position order (written by this wavefront) and `own` is this lane's
// 8 bytes raw[8*lane .. 8*lane+8), little-endian.  n: valid bytes (1..512); last: the frame's last tile (BFINAL).
// Leaves the complete image of the tile's share of the deflate stream in L.out[0 .. size) and returns size (<= n + 5, wave-uniform).
//
// A sequence is [literals][match]; in a sparse map almost every one is at most two literals and one match of at most 258 bytes: the lane
// builds those as two 32-bit strings while it sizes them - LIT (the block header on sequence 0, up to two literals: <= 21 bits) and
// M1 (<= 25 bits) -, the wave scan of the sizes places them, three LDS ORs write them.  What does not fit - the second half of a match
// of more than 258 bytes (M2), literal runs of three and more (a loop over the run through a BitSink) - goes beside them, in branches the
// wave only enters when one of its lanes needs them.  Rounds of 64 sequences are sized and written one after the other (nothing is kept
// across rounds); a tile that turns out not to shrink is rewritten as a stored block.
template <bool EVENTS = true>
__device__ __forceinline__ uint32_t deflate_encode_block(uint64_t own, uint32_t n, bool last, Lz4Lds &L)
{
    static_assert(sizeof(L.out) >= TILE_BM + 5 + 16 && sizeof(L.out) % 16 == 0, "the image: a stored block, and three dwords behind the last code");
    const int lane = lane_id();
    uint32_t *const out32 = reinterpret_cast<uint32_t *>(L.out);
    LZ4_PH_BEGIN
    // ---- phase 1: the parse (rc_lz4_block.h) ---------------------------------------------------------------------------------
    uint32_t nm = 0xFFFFFFFFu;
    bool off1 = true, store = false;
    if (EVENTS) {
        uint32_t nev = 0;
        nm = lz4_parse_events(own, n, L, nev);
        off1 = nm == 0xFFFFFFFFu;
        store = nev > (uint32_t)LZ4_NZ_STORE;        // hopeless (see LZ4_NZ_STORE): stored without a parse
    }
    if (!store && nm == 0xFFFFFFFFu) nm = lz4_parse_runs(own, n, L);
    LZ4_PH(0);
    // the image: zeroed, then ORed together (the parsers' event list in L.out is dead by now)
    if (lane < (int)(sizeof(L.out) / 16)) reinterpret_cast<u32x4 *>(L.out)[lane] = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();

    // ---- phase 2: one sequence per lane, sized and written round by round ---------------------------------------------------------
    uint32_t carry = 0;
    const uint32_t nrounds = store ? 0u : (nm + 64) / 64;
    const uint32_t *raw32 = reinterpret_cast<const uint32_t *>(L.raw);
    const uint32_t hdr = (last ? 1u : 0u) | 2u;      // BFINAL, BTYPE 01
#pragma unroll 1
    for (uint32_t rd = 0; rd < nrounds; ++rd) {
        const uint32_t k = rd * 64 + lane;
        const bool act = k <= nm;
        const uint32_t fs = L.fl[k], msk = L.ms[k], fnext = L.fl[k + 1];       // (entries behind the tables' ends: never used)
        const uint32_t off = EVENTS && !off1 ? (uint32_t)L.off[k] : 1u;
        const uint32_t q = k < nm ? msk : n;
        const uint32_t ll = act ? q - fs : 0u, ml = (act && k < nm) ? fnext - q : 0u;
        const bool longlit = ll > 2u;
        const uint64_t anylong = __builtin_amdgcn_ballot_w64(longlit);
        uint32_t lit = 0, nl = 0, pre = 0;        // LIT and its bits; bits in front of it (a long literal run and its header)
        if (k == 0 && !longlit) { lit = hdr; nl = 3; }
        if (ll && !longlit) {
            const uint32_t a = fs >> 2;
            const uint32_t v = __builtin_amdgcn_alignbyte(raw32[a + 1], raw32[a], fs & 3u);
            uint32_t len;
            lit |= deflate_lit(v & 0xFFu, len) << nl;
            nl += len;
            if (ll == 2u) {
                lit |= deflate_lit((v >> 8) & 0xFFu, len) << nl;
                nl += len;
            }
        }
        if (anylong) {
            if (longlit) pre = (k == 0 ? 3u : 0u) + 8u * ll + deflate_count_hi(L.raw, fs, ll);
        }
        const uint32_t p1 = deflate_first_part(ml);
        uint32_t m1 = 0, n1 = 0, m2 = 0, n2 = 0;
        if (ml) m1 = deflate_match(p1, off, n1);
        const uint64_t anysplit = __builtin_amdgcn_ballot_w64(ml > 258u);
        if (anysplit) {
            if (ml > 258u) m2 = deflate_match(ml - p1, off, n2);
        }
        const uint32_t bits = pre + nl + n1 + n2;
        LZ4_PH(1);
        const uint32_t sinc = wave_incl_scan(bits);
        const uint32_t o = carry + sinc - bits;
        carry += wave_last(sinc);
        if (anylong) {
            if (longlit) {
                BitSink s = sink_at(out32, o);
                if (k == 0) sink_put(s, hdr, 3);
                for (uint32_t i = 0; i < ll; i += 4) {
                    const uint32_t a = (fs + i) >> 2;
                    const uint32_t v = __builtin_amdgcn_alignbyte(raw32[a + 1], raw32[a], (fs + i) & 3u);
                    const uint32_t rem = ll - i;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        if (j < rem) {
                            uint32_t len;
                            const uint32_t code = deflate_lit((v >> (8 * j)) & 0xFFu, len);
                            sink_put(s, code, len);
                        }
                    }
                }
                sink_flush(s);
            }
        }
        // LIT and M1 as one string of <= 46 bits (a lane without a sequence: zero bits at the round's end - an OR of nothing)
        const uint64_t u = (uint64_t)lit | ((uint64_t)m1 << nl);
        or_bits64(out32, o + pre, (uint32_t)u, (uint32_t)(u >> 32));
        if (anysplit) or_bits32(out32, o + pre + nl + n1, m2);
        LZ4_PH(2);
    }
    const uint32_t T = carry + 7u;                         // ... and the end-of-block code (seven zero bits)
    uint32_t size = last ? (T + 7u) >> 3 : ((T + 3u + 7u) >> 3) + 4u;
    if (store || size >= n + 5u) {
        // stored block: [BFINAL, BTYPE 00, padding][LEN][NLEN][the bytes]
        if (!store) {   // (a parse that did not pay: the image is cleared again)
            __builtin_amdgcn_wave_barrier();
            if (lane < (int)(sizeof(L.out) / 16)) reinterpret_cast<u32x4 *>(L.out)[lane] = u32x4{0u, 0u, 0u, 0u};
            __builtin_amdgcn_wave_barrier();
        }
        size = n + 5u;
        if (lane == 0) or_bits64(out32, 0, (last ? 1u : 0u) | (n << 8) | ((~n & 0xFFu) << 24), (~n >> 8) & 0xFFu);
        if ((uint32_t)(8 * lane) < n) or_bits64(out32, 8u * (5u + 8u * (uint32_t)lane), (uint32_t)own, (uint32_t)(own >> 32));
        __builtin_amdgcn_wave_barrier();
        return size;
    }
    // the empty stored block behind the end-of-block code: only its NLEN has set bits
    if (!last && lane == 63) or_bits32(out32, 8u * (size - 2u), 0xFFFFu);
    __builtin_amdgcn_wave_barrier();
    LZ4_PH(3);
    return size;
}

}  // namespace rc
