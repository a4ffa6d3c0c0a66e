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
// (rc_deflate.hip::k_zlib_trailers).  Wave-collective; `own` = the lane's 8 bytes (bytes behind the map's end are zero).
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

// Wave-collective.  Precondition: L.raw holds the block in position order (written by this wavefront) and `own` is this lane's
// 8 bytes raw[8*lane .. 8*lane+8), little-endian.  n: valid bytes (1..512); last: the frame's last tile (BFINAL).
// Leaves the complete image of the tile's share of the deflate stream in L.out[0 .. size) and returns size (<= n + 5, wave-uniform).
template <bool EVENTS = true>
__device__ __forceinline__ uint32_t deflate_encode_block(uint64_t own, uint32_t n, bool last, Lz4Lds &L)
{
    const int lane = lane_id();
    uint32_t *const out32 = reinterpret_cast<uint32_t *>(L.out);
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

    // ---- phase 2: one sequence per lane - sizes in bits ---------------------------------------------------------------------------
    uint32_t carry = 3;                                    // the block header
    uint32_t seq_o[2], seq_fs[2], seq_ll[2], seq_ml[2], seq_off[2];   // at most 2 rounds of 64 sequences
    const uint32_t nrounds = store ? 0u : (nm + 64) / 64;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        seq_ll[rd] = 0xFFFFFFFFu;  // "no sequence"
        if ((uint32_t)rd < nrounds) {
            const uint32_t k = rd * 64 + lane;
            uint32_t bits = 0;
            if (k <= nm) {
                const uint32_t fs = L.fl[k], msk = L.ms[k], fnext = L.fl[k + 1];
                const uint32_t off = EVENTS && !off1 ? (uint32_t)L.off[k] : 1u;
                const uint32_t q = k < nm ? msk : n;
                const uint32_t ll = q - fs, ml = k < nm ? fnext - q : 0u;
                bits = 8u * ll + deflate_count_hi(L.raw, fs, ll);
                if (ml) {
                    uint32_t len;
                    const uint32_t p1 = deflate_first_part(ml);
                    (void)deflate_match(p1, off, len);
                    bits += len;
                    if (p1 != ml) { (void)deflate_match(ml - p1, off, len); bits += len; }
                }
                seq_fs[rd] = fs; seq_ll[rd] = ll; seq_ml[rd] = ml; seq_off[rd] = off;
            }
            const uint32_t sinc = wave_incl_scan(bits);
            seq_o[rd] = carry + sinc - bits;
            carry += wave_last(sinc);
        }
    }
    const uint32_t T = carry + 7u;                         // ... and the end-of-block code (seven zero bits)
    uint32_t size = last ? (T + 7u) >> 3 : ((T + 3u + 7u) >> 3) + 4u;
    if (store || size >= n + 5u) { store = true; size = n + 5u; }

    // ---- the image: zeroed, then ORed together ---------------------------------------------------------------------------------------
    if (lane < (int)(sizeof(L.out) / 16)) reinterpret_cast<u32x4 *>(L.out)[lane] = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    if (store) {
        // stored block: [BFINAL, BTYPE 00, padding][LEN][NLEN][the bytes]
        if (lane == 0) {
            BitSink s = sink_at(out32, 0);
            sink_put(s, last ? 1u : 0u, 8);
            sink_put(s, n | ((~n & 0xFFFFu) << 16), 32);
            sink_flush(s);
        }
        if ((uint32_t)(8 * lane) < n) {
            BitSink s = sink_at(out32, 8u * (5u + 8u * (uint32_t)lane));
            sink_put(s, (uint32_t)own, 32);
            sink_put(s, (uint32_t)(own >> 32), 32);
            sink_flush(s);
        }
        __builtin_amdgcn_wave_barrier();
        return size;
    }
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        if ((uint32_t)rd < nrounds && seq_ll[rd] != 0xFFFFFFFFu) {
            const uint32_t ll = seq_ll[rd], fs = seq_fs[rd], ml = seq_ml[rd], off = seq_off[rd];
            const bool first = rd == 0 && lane == 0;
            BitSink s = sink_at(out32, first ? 0u : seq_o[rd]);
            if (first) sink_put(s, (last ? 1u : 0u) | 2u, 3);     // BFINAL, BTYPE 01
            const uint32_t *raw32 = reinterpret_cast<const uint32_t *>(L.raw);
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
            if (ml) {
                uint32_t len;
                const uint32_t p1 = deflate_first_part(ml);
                uint32_t code = deflate_match(p1, off, len);
                sink_put(s, code, len);
                if (p1 != ml) { code = deflate_match(ml - p1, off, len); sink_put(s, code, len); }
            }
            sink_flush(s);
        }
    }
    // the empty stored block behind the end-of-block code: only its NLEN has set bits
    if (!last && lane == 63) {
        BitSink s = sink_at(out32, 8u * (size - 2u));
        sink_put(s, 0xFFFFu, 16);
        sink_flush(s);
    }
    __builtin_amdgcn_wave_barrier();
    return size;
}

}  // namespace rc
