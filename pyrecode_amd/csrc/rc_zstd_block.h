// rc_zstd_block.h - Zstandard block encoder for packed binary maps (one 512-byte tile per block), written from the format
// specification (RFC 8878).  Shared by the HIP kernel (one LANE encodes one block) and by a host-only checker build
// (tests/zstd_host_check.cpp) that feeds the same bytes to the stock libzstd decoder.
//
// Replaces the reference's `ZstdCompressor(level, write_content_size=False).compress(data)` on the packed binary map
// (pyrecode/recode_writer.py:175-178, recode_compressors.py:88, called from recode_writer.py:503-505).  The reference pins
// no compressed bytes for this scheme: the contract is a valid zstd frame that the stock decoder expands to
// the bit-exact input.
//
// Block encoding (sparse bitmaps: > 90 % zero bytes at the target sparsity):
//   all-zero tile        -> RLE block (4 bytes)
//   otherwise            -> Compressed block: Raw literals + sequences; every run of >= 4 zero bytes is
//                           [literal 0x00][match: repeat-offset 1, length run-1] (an overlapping copy = RLE).
//                           Literal lengths and match lengths use the PREDEFINED FSE tables, offsets use RLE mode with
//                           code 0 (= "repeat offset 1", zero bits per sequence; the frame starts with rep1 = 1 and only
//                           offset 1 is ever used, and every sequence has >= 1 literal so the code never shifts).
//   would not shrink     -> Raw block
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define RC_HD __host__ __device__ __forceinline__
#else
#define RC_HD inline
#endif

namespace rc {

constexpr int ZSTD_BLK = 512;
constexpr int ZSTD_SLOT_MAX = ZSTD_BLK + 3;  // raw block: 3-byte header + payload

// FSE compression tables of a literal-length and a match-length distribution: the format's predefined ones (accuracy log 6,
// zstd_build_tables) or a ctx's fitted ones (accuracy log up to 9, rc_zstd_model.h).
struct ZstdTables {
    uint16_t ll_state[512], ml_state[512];
    uint32_t ll_dnb[36], ml_dnb[53];   // deltaNbBits
    int32_t ll_dfs[36], ml_dfs[53];    // deltaFindState
    uint32_t ll_log, ml_log;           // accuracy logs
};

// ---- the "modelled" encoder's tables (fitted on the host, rc_zstd_model.h; used by the kernels) -----------------------------
constexpr int ZM_HUF_MAXBITS = 11;   // Huffman: the format's maximum code length
constexpr int ZM_LL_SYMS = 36, ZM_ML_SYMS = 53;
constexpr int ZM_DESC_MAX = 192;     // bytes reserved for one description
// a flat POD copied to the device as it stands
struct ZstdModel {
    // Huffman code of every byte value: code value | length << 12 (length 1..11), for the bitmap literals and for the bytes of
    // the packed residual stream
    uint16_t lit_code[256];
    uint16_t pix_code[256];
    ZstdTables seq;                       // FSE compression tables of the literal-length / match-length codes
    uint8_t lit_desc[ZM_DESC_MAX];        // Huffman tree description of lit_code (as it stands in a Compressed_Literals_Block)
    uint8_t pix_desc[ZM_DESC_MAX];
    uint8_t seq_desc[ZM_DESC_MAX];        // [LL FSE table description][0x00 = the offset code of RLE mode][ML description]
    uint32_t lit_desc_len, pix_desc_len, seq_desc_len;
    uint32_t valid;                       // bit 0: lit, bit 1: pix, bit 2: seq usable; bit 3: LITERALS ONLY - the binary maps' blocks carry all
                                          // their bytes as Huffman-coded literals and no sequences (dense maps: zm_build_model), lit_code
                                          // is then fitted to ALL bytes of the non-empty blocks
};
constexpr uint32_t ZM_LITS_ONLY = 8u;
struct ZstdSample {   // histograms gathered by k_zstd_sample over a sample of frames
    uint32_t lit[256], pix[256], ll[64], ml[64];
    uint32_t all[256];                    // every byte of the sample's binary-map blocks that are not all zero (those are RLE blocks either way)
    uint32_t nblk, pad[3];                // how many such blocks
};

// (host side) FSE_buildCTable of the reference implementation, restated: spread symbols with step (size/2 + size/8 + 3), low-probability
// (-1) symbols from the top of the table down; state table sorted by symbol; per-symbol transform.
inline void zstd_build_ctable(const int16_t *norm, int nsym, int table_log, uint16_t *state_table, uint32_t *dnb, int32_t *dfs)
{
    const int size = 1 << table_log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint8_t symbol[64];
    int cumul[64 + 2];
    int high = size - 1;
    cumul[0] = 0;
    for (int u = 1; u <= nsym; ++u) {
        if (norm[u - 1] == -1) {
            cumul[u] = cumul[u - 1] + 1;
            symbol[high--] = (uint8_t)(u - 1);
        } else {
            cumul[u] = cumul[u - 1] + norm[u - 1];
        }
    }
    cumul[nsym] = size + 1;
    int pos = 0;
    for (int s = 0; s < nsym; ++s)
        for (int i = 0; i < norm[s]; ++i) {
            symbol[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    {
        int c2[64 + 2];
        for (int i = 0; i <= nsym; ++i) c2[i] = cumul[i];
        for (int u = 0; u < size; ++u) state_table[c2[symbol[u]]++] = (uint16_t)(size + u);
    }
    int total = 0;
    for (int s = 0; s < nsym; ++s) {
        const int n = norm[s];
        if (n == 0) {
            dnb[s] = (uint32_t)(((table_log + 1) << 16) - (1 << table_log));
            dfs[s] = 0;
        } else if (n == -1 || n == 1) {
            dnb[s] = (uint32_t)((table_log << 16) - (1 << table_log));
            dfs[s] = total - 1;
            total++;
        } else {
            int hb = 31 - __builtin_clz((unsigned)(n - 1));
            const int max_bits = table_log - hb;
            const int min_state_plus = n << max_bits;
            dnb[s] = (uint32_t)((max_bits << 16) - min_state_plus);
            dfs[s] = total - n;
            total += n;
        }
    }
}
inline void zstd_build_tables(ZstdTables &t)
{
    static const int16_t ll[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
    static const int16_t ml[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
    zstd_build_ctable(ll, 36, 6, t.ll_state, t.ll_dnb, t.ll_dfs);
    zstd_build_ctable(ml, 53, 6, t.ml_state, t.ml_dnb, t.ml_dfs);
    t.ll_log = 6;
    t.ml_log = 6;
}

// literal-length / match-length value -> (code, number of extra bits); the extra bits are value - baseline(code)
RC_HD void zstd_ll_code(uint32_t ll, uint32_t &code, uint32_t &nbits, uint32_t &extra)
{
    if (ll < 16) { code = ll; nbits = 0; extra = 0; return; }
    if (ll < 24) { code = 16 + ((ll - 16) >> 1); nbits = 1; extra = (ll - 16) & 1; return; }
    if (ll < 32) { code = 20 + ((ll - 24) >> 2); nbits = 2; extra = (ll - 24) & 3; return; }
    if (ll < 48) { code = 22 + ((ll - 32) >> 3); nbits = 3; extra = (ll - 32) & 7; return; }
    if (ll < 64) { code = 24; nbits = 4; extra = ll - 48; return; }
    if (ll < 128) { code = 25; nbits = 6; extra = ll - 64; return; }
    if (ll < 256) { code = 26; nbits = 7; extra = ll - 128; return; }
    if (ll < 512) { code = 27; nbits = 8; extra = ll - 256; return; }
    code = 28; nbits = 9; extra = ll - 512;  // <= 1023; a block holds at most 512 literals
}
RC_HD void zstd_ml_code(uint32_t ml, uint32_t &code, uint32_t &nbits, uint32_t &extra)
{
    const uint32_t b = ml - 3;  // match length >= 3
    if (b < 32) { code = b; nbits = 0; extra = 0; return; }
    if (b < 40) { code = 32 + ((b - 32) >> 1); nbits = 1; extra = (b - 32) & 1; return; }
    if (b < 48) { code = 36 + ((b - 40) >> 2); nbits = 2; extra = (b - 40) & 3; return; }
    if (b < 64) { code = 38 + ((b - 48) >> 3); nbits = 3; extra = (b - 48) & 7; return; }
    if (b < 96) { code = 40 + ((b - 64) >> 4); nbits = 4; extra = (b - 64) & 15; return; }
    if (b < 128) { code = 42; nbits = 5; extra = b - 96; return; }
    if (b < 256) { code = 43; nbits = 7; extra = b - 128; return; }
    code = 44; nbits = 8; extra = b - 256;  // match length <= 514
}

// backward bit writer of the sequences section: bits accumulate LSB-first, bytes are emitted little-endian
struct ZstdBits {
    uint64_t acc;
    uint32_t n;
    uint8_t *p;
};
RC_HD void zb_add(ZstdBits &b, uint32_t value, uint32_t nbits)
{
    b.acc |= (uint64_t)(value & ((1u << nbits) - 1u)) << b.n;
    b.n += nbits;
}
RC_HD void zb_flush(ZstdBits &b)
{
    while (b.n >= 8) {
        *b.p++ = (uint8_t)b.acc;
        b.acc >>= 8;
        b.n -= 8;
    }
}

struct ZstdSeq { uint16_t ll, ml; };

// Encode one block: src[0..n) -> dst (capacity ZSTD_SLOT_MAX + 8).  `seq` is caller scratch for up to n/4 + 1 sequences.
// Returns the number of bytes written (block header included).  `last` sets Last_Block.
RC_HD uint32_t zstd_encode_block(const uint8_t *src, uint32_t n, uint8_t *dst, ZstdSeq *seq, const ZstdTables &T, bool last)
{
    const uint32_t lastbit = last ? 1u : 0u;
    // pass 1: sequences = (literal run, zero run >= 4 minus its first byte); literals are everything not matched
    uint32_t nseq = 0, nlit = 0, any = 0;
    {
        uint32_t i = 0, lit_start = 0;
        while (i < n) {
            if (src[i] != 0) { any = 1; ++i; continue; }
            uint32_t j = i + 1;
            while (j < n && src[j] == 0) ++j;
            const uint32_t run = j - i;
            if (run >= 4) {  // literal zero at i, match of run-1 >= 3 bytes
                seq[nseq].ll = (uint16_t)(i + 1 - lit_start);
                seq[nseq].ml = (uint16_t)(run - 1);
                nlit += i + 1 - lit_start;
                ++nseq;
                lit_start = j;
            }
            i = j;
        }
        nlit += n - lit_start;  // trailing literals (not part of any sequence)
    }
    if (!any) {  // RLE block: Block_Size = regenerated size, one byte of content
        const uint32_t h = lastbit | (1u << 1) | (n << 3);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
        dst[3] = 0;
        return 4;
    }
    uint8_t *p = dst + 3;
    // literals section: Raw_Literals_Block, 1-byte header for <= 31 literals else 2-byte (12-bit size)
    if (nlit < 32) {
        *p++ = (uint8_t)(nlit << 3);
    } else {
        *p++ = (uint8_t)((nlit << 4) | (1u << 2));
        *p++ = (uint8_t)(nlit >> 4);
    }
    {
        uint32_t i = 0;
        for (uint32_t s = 0; s < nseq; ++s) {
            for (uint32_t k = 0; k < seq[s].ll; ++k) *p++ = src[i + k];
            i += seq[s].ll + seq[s].ml;
        }
        while (i < n) *p++ = src[i++];
    }
    // sequences section
    if (nseq < 128) {
        *p++ = (uint8_t)nseq;
    } else {
        *p++ = (uint8_t)(128 + (nseq >> 8));
        *p++ = (uint8_t)nseq;
    }
    if (nseq) {
        *p++ = (uint8_t)(1u << 4);  // LL predefined (0), OF RLE (1), ML predefined (0)
        *p++ = 0;                   // the single offset code: 0 = repeat offset 1, no extra bits
        ZstdBits b{0, 0, p};
        uint32_t llc, llb, lle, mlc, mlb, mle;
        zstd_ll_code(seq[nseq - 1].ll, llc, llb, lle);
        zstd_ml_code(seq[nseq - 1].ml, mlc, mlb, mle);
        // initial states from the LAST sequence (FSE_initCState2)
        uint32_t st_ml, st_ll;
        {
            const uint32_t nb = (T.ml_dnb[mlc] + (1u << 15)) >> 16;
            const uint32_t v = (nb << 16) - T.ml_dnb[mlc];
            st_ml = T.ml_state[(int32_t)(v >> nb) + T.ml_dfs[mlc]];
        }
        {
            const uint32_t nb = (T.ll_dnb[llc] + (1u << 15)) >> 16;
            const uint32_t v = (nb << 16) - T.ll_dnb[llc];
            st_ll = T.ll_state[(int32_t)(v >> nb) + T.ll_dfs[llc]];
        }
        zb_add(b, lle, llb);
        zb_add(b, mle, mlb);
        zb_flush(b);  // (offset: 0 extra bits)
        for (uint32_t s = nseq - 1; s-- > 0;) {
            zstd_ll_code(seq[s].ll, llc, llb, lle);
            zstd_ml_code(seq[s].ml, mlc, mlb, mle);
            {   // FSE_encodeSymbol: offset state has 0 bits (RLE), then match length, then literal length
                const uint32_t nb = (st_ml + T.ml_dnb[mlc]) >> 16;
                zb_add(b, st_ml, nb);
                st_ml = T.ml_state[(int32_t)(st_ml >> nb) + T.ml_dfs[mlc]];
            }
            {
                const uint32_t nb = (st_ll + T.ll_dnb[llc]) >> 16;
                zb_add(b, st_ll, nb);
                st_ll = T.ll_state[(int32_t)(st_ll >> nb) + T.ll_dfs[llc]];
            }
            zb_flush(b);
            zb_add(b, lle, llb);
            zb_add(b, mle, mlb);
            zb_flush(b);
        }
        zb_add(b, st_ml, T.ml_log);  // FSE_flushCState: match length, (offset: 0 bits), literal length
        zb_add(b, st_ll, T.ll_log);
        zb_add(b, 1, 1);      // end mark
        zb_flush(b);
        if (b.n) { *b.p++ = (uint8_t)b.acc; }
        p = b.p;
    }
    const uint32_t content = (uint32_t)(p - (dst + 3));
    if (content >= n) {  // would not shrink: Raw block
        const uint32_t h = lastbit | (0u << 1) | (n << 3);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
        for (uint32_t i = 0; i < n; ++i) dst[3 + i] = src[i];
        return 3 + n;
    }
    const uint32_t h = lastbit | (2u << 1) | (content << 3);
    dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
    return 3 + content;
}

// ---- token form: the split the GPU runs (rc_zstd_wave.h) ------------------------------------------------------------------
// The byte-parallel half of the encoder leaves, per block, the fixed part of the Compressed block followed by one 32-bit
// token per sequence, LAST sequence of the block first:  llc | mlc << 6 | ll_extra << 12 | ml_extra << 21.
// fse_chain below is the serial half (shared by the HIP kernel k_zstd_fse and the host-side format check).
constexpr uint32_t ZW_FINAL = 0x80000000u;  // blk_size word: the slot already holds a complete block of (word & 0xFFFF) bytes
// otherwise: P | nseq << 16  (P = offset of the FSE bitstream, tokens at zstd_token_offset(P))
RC_HD uint32_t zstd_token_offset(uint32_t P) { return (P + 15u) & ~15u; }  // tokens are fetched by 16-byte loads

// number of extra bits of a literal-length / match-length code: 13 nibbles for codes 16.. / 32..
RC_HD uint32_t zstd_ll_bits(uint32_t c) { return c < 16 ? 0u : (uint32_t)(0x9876433221111ull >> (4 * (c - 16))) & 15u; }
RC_HD uint32_t zstd_ml_bits(uint32_t c) { return c < 32 ? 0u : (uint32_t)(0x8754433221111ull >> (4 * (c - 32))) & 15u; }

struct alignas(16) ZW4 { uint32_t v[4]; };
struct FseState { uint32_t st_ml, st_ll; };
RC_HD void fse_first(FseState &f, uint32_t llc, uint32_t mlc, const ZstdTables &T)
{   // FSE_initCState2
    uint32_t b = (T.ml_dnb[mlc] + (1u << 15)) >> 16;
    uint32_t v = (b << 16) - T.ml_dnb[mlc];
    f.st_ml = T.ml_state[(int32_t)(v >> b) + T.ml_dfs[mlc]];
    b = (T.ll_dnb[llc] + (1u << 15)) >> 16;
    v = (b << 16) - T.ll_dnb[llc];
    f.st_ll = T.ll_state[(int32_t)(v >> b) + T.ll_dfs[llc]];
}

// ---- modelled blocks: Huffman-coded literals + fitted FSE tables, defined once per frame ----------------------------------
// Every block is encoded as if the frame's tables were already known to the decoder (Treeless_Literals_Block, Repeat_Mode for
// all three sequence tables); the FIRST block of a frame that uses the Huffman tree / the sequence tables then gets their
// descriptions inserted (zm_insert_defs, done per frame behind the block encoders).  These serial forms are the
// specification of what the wave-collective encoders produce (rc_zstd_wave.h, rc_pix_huff.hip) and what the CPU format
// check feeds to stock libzstd.

// Single-stream Huffman bitstream of lits[0..nlit): the LAST literal occupies the lowest bits, the first literal the
// highest, then the end mark.  Returns the stream's byte count (dst needs nlit * 11 / 8 + 2 bytes).
RC_HD uint32_t zm_huf_stream(const uint8_t *lits, uint32_t nlit, const uint16_t *code, uint8_t *dst)
{
    uint64_t acc = 0;
    uint32_t nb = 0, o = 0;
    for (uint32_t i = nlit; i-- > 0;) {
        const uint32_t c = code[lits[i]];
        acc |= (uint64_t)(c & 0xFFFu) << nb;
        nb += c >> 12;
        while (nb >= 8) { dst[o++] = (uint8_t)acc; acc >>= 8; nb -= 8; }
    }
    acc |= (uint64_t)1 << nb;
    ++nb;
    while (nb > 0) { dst[o++] = (uint8_t)acc; acc >>= 8; nb = nb > 8 ? nb - 8 : 0; }
    return o;
}
// 3-byte literals section header, size format 0 (single stream, 10-bit sizes): type 2 = with tree, 3 = treeless
RC_HD void zm_lit_header(uint8_t *p, uint32_t type, uint32_t regen, uint32_t comp)
{
    const uint32_t h = type | (regen << 4) | (comp << 14);
    p[0] = (uint8_t)h; p[1] = (uint8_t)(h >> 8); p[2] = (uint8_t)(h >> 16);
}
RC_HD uint32_t zm_raw_lit_header(uint8_t *p, uint32_t nlit)
{
    if (nlit < 32) { p[0] = (uint8_t)(nlit << 3); return 1; }
    p[0] = (uint8_t)((nlit << 4) | (1u << 2)); p[1] = (uint8_t)(nlit >> 4);   // 12-bit size (nlit <= 4095)
    return 2;
}
// a compressed block may take part only if it still fits its slot once BOTH descriptions have been inserted into it
RC_HD uint32_t zm_block_budget(const ZstdModel &M, uint32_t slot_bytes)
{
    return slot_bytes - 8u - ((M.valid & 1u) ? M.lit_desc_len : 0u) - ((M.valid & 4u) ? M.seq_desc_len : 0u);
}

// One block, serial (host only): the same parse as zstd_encode_block (zero runs >= 4 -> [literal 00][match: repeat offset 1]).
// dst capacity: ZSTD_SLOT_MAX + 8.  lits / hbuf: caller scratch of n and n * 11 / 8 + 2 bytes.
inline uint32_t zstd_encode_block_model(const uint8_t *src, uint32_t n, uint8_t *dst, ZstdSeq *seq, uint8_t *lits, uint8_t *hbuf,
                                       const ZstdModel &M, bool last, uint32_t slot_bytes = 640)
{
    const uint32_t lastbit = last ? 1u : 0u;
    uint32_t nseq = 0, nlit = 0, any = 0;
    {
        uint32_t i = 0, lit_start = 0;
        while (i < n) {
            if (src[i] != 0) { any = 1; ++i; continue; }
            uint32_t j = i + 1;
            while (j < n && src[j] == 0) ++j;
            if (j - i >= 4) {
                for (uint32_t k = lit_start; k <= i; ++k) lits[nlit++] = src[k];
                seq[nseq].ll = (uint16_t)(i + 1 - lit_start);
                seq[nseq].ml = (uint16_t)(j - i - 1);
                ++nseq;
                lit_start = j;
            }
            i = j;
        }
        for (uint32_t k = lit_start; k < n; ++k) lits[nlit++] = src[k];
    }
    if (!any) {
        const uint32_t h = lastbit | (1u << 1) | (n << 3);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); dst[3] = 0;
        return 4;
    }
    if (M.valid & ZM_LITS_ONLY) {   // dense maps: every byte a literal, no sequences (what zstd_tokenize_block_m emits under this model)
        nseq = 0;
        nlit = n;
        for (uint32_t k = 0; k < n; ++k) lits[k] = src[k];
    }
    uint8_t *p = dst + 3;
    uint32_t hb = 0;
    if (M.valid & 1u) hb = zm_huf_stream(lits, nlit, M.lit_code, hbuf);
    if ((M.valid & 1u) && 3 + hb < zm_raw_lit_header(p, nlit) + nlit) {   // Huffman pays
        zm_lit_header(p, 3, nlit, hb);
        p += 3;
        for (uint32_t i = 0; i < hb; ++i) *p++ = hbuf[i];
    } else {
        p += zm_raw_lit_header(p, nlit);
        for (uint32_t i = 0; i < nlit; ++i) *p++ = lits[i];
    }
    if (nseq < 128) *p++ = (uint8_t)nseq;
    else { *p++ = (uint8_t)(128 + (nseq >> 8)); *p++ = (uint8_t)nseq; }
    if (nseq) {
        const bool fitted = (M.valid & 4u) != 0;
        ZstdTables pre;
        if (!fitted) zstd_build_tables(pre);
        const ZstdTables &T = fitted ? M.seq : pre;
        if (fitted) *p++ = 0xFC;                       // Repeat_Mode x 3
        else { *p++ = (uint8_t)(1u << 4); *p++ = 0; }  // predefined, RLE offsets (code 0), predefined
        ZstdBits b{0, 0, p};
        uint32_t llc, llb, lle, mlc, mlb, mle;
        zstd_ll_code(seq[nseq - 1].ll, llc, llb, lle);
        zstd_ml_code(seq[nseq - 1].ml, mlc, mlb, mle);
        FseState f;
        fse_first(f, llc, mlc, T);
        zb_add(b, lle, llb);
        zb_add(b, mle, mlb);
        zb_flush(b);
        for (uint32_t s = nseq - 1; s-- > 0;) {
            zstd_ll_code(seq[s].ll, llc, llb, lle);
            zstd_ml_code(seq[s].ml, mlc, mlb, mle);
            uint32_t nb = (f.st_ml + T.ml_dnb[mlc]) >> 16;
            zb_add(b, f.st_ml, nb);
            f.st_ml = T.ml_state[(int32_t)(f.st_ml >> nb) + T.ml_dfs[mlc]];
            nb = (f.st_ll + T.ll_dnb[llc]) >> 16;
            zb_add(b, f.st_ll, nb);
            f.st_ll = T.ll_state[(int32_t)(f.st_ll >> nb) + T.ll_dfs[llc]];
            zb_flush(b);
            zb_add(b, lle, llb);
            zb_add(b, mle, mlb);
            zb_flush(b);
        }
        zb_add(b, f.st_ml, T.ml_log);
        zb_add(b, f.st_ll, T.ll_log);
        zb_add(b, 1, 1);
        zb_flush(b);
        if (b.n) *b.p++ = (uint8_t)b.acc;
        p = b.p;
    }
    const uint32_t content = (uint32_t)(p - (dst + 3));
    if (content >= n || 3 + content > zm_block_budget(M, slot_bytes)) {   // would not shrink (or no room for the descriptions): Raw block
        const uint32_t h = lastbit | (n << 3);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
        for (uint32_t i = 0; i < n; ++i) dst[3 + i] = src[i];
        return 3 + n;
    }
    const uint32_t h = lastbit | (2u << 1) | (content << 3);
    dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
    return 3 + content;
}

// What a finished block needs from the frame's definitions: bit 0 = it is a compressed block with treeless literals,
// bit 1 = it is a compressed block whose sequences use Repeat_Mode.  *seq_pos = offset just behind its modes byte.
RC_HD uint32_t zm_block_needs(const uint8_t *blk, uint32_t *seq_pos)
{
    *seq_pos = 0;
    if (((blk[0] >> 1) & 3u) != 2u) return 0;
    const uint32_t lt = blk[3] & 3u;
    uint32_t needs = 0, lsz;   // lsz: literals section bytes (header + content)
    if (lt == 3u || lt == 2u) {
        const uint32_t h = (uint32_t)blk[3] | ((uint32_t)blk[4] << 8) | ((uint32_t)blk[5] << 16);
        lsz = 3 + (h >> 14);
        if (lt == 3u) needs |= 1u;
    } else {
        const uint32_t sf = (blk[3] >> 2) & 3u;
        lsz = (sf == 1u) ? 2 + (((uint32_t)blk[3] >> 4) | ((uint32_t)blk[4] << 4)) : 1 + (blk[3] >> 3);
    }
    const uint32_t q = 3 + lsz;
    const uint32_t nb = blk[q] < 128 ? 1u : 2u;
    if (blk[q] != 0 && blk[q + nb] == 0xFC) { needs |= 2u; *seq_pos = q + nb + 1; }
    return needs;
}
// Byte i of the block with the descriptions selected by `add` inserted (tree behind the literals header at offset 6, sequence
// tables behind the modes byte at seq_pos) and the three affected header fields patched.  Pure function of i: the device
// applies it with one thread per byte, the host check in a loop.  tl / sl: lengths to insert (0 = not this block); the new
// block is tl + sl bytes longer.
RC_HD uint8_t zm_defs_byte(const uint8_t *blk, uint32_t seq_pos, const uint8_t *tree, uint32_t tl, const uint8_t *sdesc, uint32_t sl,
                           uint32_t i)
{
    if (i < 3) {   // Block_Header: Block_Size grows
        const uint32_t h = ((uint32_t)blk[0] | ((uint32_t)blk[1] << 8) | ((uint32_t)blk[2] << 16)) + ((tl + sl) << 3);
        return (uint8_t)(h >> (8 * i));
    }
    if (i < 6 && tl) {   // literals header: treeless -> with tree, compressed size grows
        uint32_t h = (uint32_t)blk[3] | ((uint32_t)blk[4] << 8) | ((uint32_t)blk[5] << 16);
        h = (h & ~3u) | 2u;
        h += tl << 14;
        return (uint8_t)(h >> (8 * (i - 3)));
    }
    if (tl && i >= 6 && i < 6 + tl) return tree[i - 6];
    uint32_t j = i - tl;                 // position in the original block (behind the tree insertion)
    if (sl) {
        if (j == seq_pos - 1) return 0x98;   // modes: FSE_Compressed, RLE, FSE_Compressed
        if (j >= seq_pos) {
            if (j < seq_pos + sl) return sdesc[j - seq_pos];
            j -= sl;
        }
    }
    return blk[j];
}

// One chunk of the packed residual stream as a block of Huffman-coded literals without sequences, or a Raw block.
// dst capacity n + 16; hbuf scratch n * 11 / 8 + 2.
RC_HD uint32_t zm_encode_pix_chunk(const uint8_t *src, uint32_t n, uint8_t *dst, uint8_t *hbuf, const ZstdModel &M, bool last)
{
    const uint32_t lastbit = last ? 1u : 0u;
    uint32_t hb = 0;
    if ((M.valid & 2u) && n) hb = zm_huf_stream(src, n, M.pix_code, hbuf);
    const uint32_t budget = n > M.pix_desc_len + 8u ? n - M.pix_desc_len - 8u : 0u;   // must still beat Raw with the tree inserted
    if (!(M.valid & 2u) || n == 0 || 3 + hb + 1 >= budget) {
        const uint32_t h = lastbit | (n << 3);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
        for (uint32_t i = 0; i < n; ++i) dst[3 + i] = src[i];
        return 3 + n;
    }
    const uint32_t content = 3 + hb + 1;
    const uint32_t h = lastbit | (2u << 1) | (content << 3);
    dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
    zm_lit_header(dst + 3, 3, n, hb);
    for (uint32_t i = 0; i < hb; ++i) dst[6 + i] = hbuf[i];
    dst[6 + hb] = 0;   // Number_of_Sequences
    return 3 + content;
}

// Runs the chain over the tokens; emit(dword) receives the finished dwords in order.  acc/nb enter holding the (P & 3)
// fixed-part bytes that share the first dword.  Returns the bit count left in acc (at most 50).  Tokens are read two
// 16-byte chunks ahead of their use, and a sequence adds at most 29 bits for its 32-bit token, so emit() may store IN PLACE
// from dword P >> 2 of the same slot: the write position never passes the read position.
template <class Emit>
RC_HD uint32_t fse_chain(const ZW4 *tok4, uint32_t nseq, uint64_t &acc, uint32_t nb, const ZstdTables &T, Emit &&emit)
{
    const uint32_t nchunk = (nseq + 3) >> 2;
    const ZW4 zero = {{0u, 0u, 0u, 0u}};
    ZW4 cur = tok4[0];
    ZW4 nxt = nchunk > 1 ? tok4[1] : zero;
    FseState f{0, 0};
    for (uint32_t c = 0; c < nchunk; ++c) {
        const ZW4 ahead = c + 2 < nchunk ? tok4[c + 2] : zero;
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = 4 * c + j;
            if (i < nseq) {
                const uint32_t t = cur.v[j];
                const uint32_t llc = t & 63u, mlc = (t >> 6) & 63u;
                if (i == 0) {
                    fse_first(f, llc, mlc, T);
                } else {  // FSE_encodeSymbol: (offset: 0 bits), match length, literal length
                    uint32_t b = (f.st_ml + T.ml_dnb[mlc]) >> 16;
                    acc |= (uint64_t)(f.st_ml & ((1u << b) - 1u)) << nb; nb += b;
                    f.st_ml = T.ml_state[(int32_t)(f.st_ml >> b) + T.ml_dfs[mlc]];
                    b = (f.st_ll + T.ll_dnb[llc]) >> 16;
                    acc |= (uint64_t)(f.st_ll & ((1u << b) - 1u)) << nb; nb += b;
                    f.st_ll = T.ll_state[(int32_t)(f.st_ll >> b) + T.ll_dfs[llc]];
                    if (nb >= 32) { emit((uint32_t)acc); acc >>= 32; nb -= 32; }
                }
                const uint32_t llb = zstd_ll_bits(llc), mlb = zstd_ml_bits(mlc);
                acc |= (uint64_t)((t >> 12) & ((1u << llb) - 1u)) << nb; nb += llb;
                acc |= (uint64_t)((t >> 21) & ((1u << mlb) - 1u)) << nb; nb += mlb;
                if (nb >= 32) { emit((uint32_t)acc); acc >>= 32; nb -= 32; }
            }
        }
        cur = nxt;
        nxt = ahead;
    }
    // FSE_flushCState: match length, (offset: 0 bits), literal length; then the end mark.  (Up to 31 bits are pending
    // here and the two states add up to 18: the caller's accumulator is 64 bits wide.)
    acc |= (uint64_t)(f.st_ml & ((1u << T.ml_log) - 1u)) << nb; nb += T.ml_log;
    acc |= (uint64_t)(f.st_ll & ((1u << T.ll_log) - 1u)) << nb; nb += T.ll_log;
    acc |= (uint64_t)1 << nb; nb += 1;
    return nb;
}

}  // namespace rc
