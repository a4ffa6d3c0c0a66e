// Host-only build of the device block encoder (pyrecode_amd/csrc/rc_zstd_block.h) for CPU tests: lets the stock libzstd
// decoder judge the bitstream the HIP kernels will emit, without a GPU.  Test infrastructure, not a product path.
//   streaming = 0: the serial restatement zstd_encode_block
//   streaming = 1: the token form the GPU runs - a scalar stand-in for the wave tokenizer (rc_zstd_wave.h) lays out the slot
//                  (fixed part, padded to 16, tokens with the last sequence first), then the SAME fse_chain the kernel
//                  k_zstd_fse runs writes the bitstream in place.
#include <cstring>
#include <vector>

#include "../../pyrecode_amd/csrc/rc_zstd_block.h"
#include "../../pyrecode_amd/csrc/rc_zstd_model.h"

// scalar stand-in for zstd_tokenize_block + k_zstd_fse on one block; slot: 640 bytes, 16-byte aligned
static uint32_t token_form_block(const uint8_t *src, uint32_t n, uint8_t *slot, const rc::ZstdTables &T, bool last)
{
    constexpr uint32_t SLOT = 640;
    std::vector<rc::ZstdSeq> seq;
    std::vector<uint8_t> lits;
    bool any = false;
    {
        uint32_t i = 0, lit_start = 0;
        while (i < n) {
            if (src[i] != 0) { any = true; ++i; continue; }
            uint32_t j = i + 1;
            while (j < n && src[j] == 0) ++j;
            if (j - i >= 4) {
                for (uint32_t k = lit_start; k <= i; ++k) lits.push_back(src[k]);
                seq.push_back({(uint16_t)(i + 1 - lit_start), (uint16_t)(j - i - 1)});
                lit_start = j;
            }
            i = j;
        }
        for (uint32_t k = lit_start; k < n; ++k) lits.push_back(src[k]);
    }
    const uint32_t lastbit = last ? 1u : 0u;
    if (!any) {
        const uint32_t h = lastbit | (1u << 1) | (n << 3);
        slot[0] = (uint8_t)h; slot[1] = (uint8_t)(h >> 8); slot[2] = (uint8_t)(h >> 16); slot[3] = 0;
        return 4;
    }
    const uint32_t nseq = (uint32_t)seq.size(), nlit = (uint32_t)lits.size();
    const uint32_t lh = nlit < 32 ? 1u : 2u, sh = nseq < 128 ? 1u : 2u;
    const uint32_t P = 3 + lh + nlit + sh + 2, T0 = rc::zstd_token_offset(P);
    bool raw = nseq == 0 || T0 + 4 * nseq + 8 > SLOT;
    if (!raw) {
        memset(slot, 0, SLOT);
        uint8_t *p = slot + 3;
        if (nlit < 32) *p++ = (uint8_t)(nlit << 3);
        else { *p++ = (uint8_t)((nlit << 4) | (1u << 2)); *p++ = (uint8_t)(nlit >> 4); }
        memcpy(p, lits.data(), nlit);
        p += nlit;
        if (nseq < 128) *p++ = (uint8_t)nseq;
        else { *p++ = (uint8_t)(128 + (nseq >> 8)); *p++ = (uint8_t)nseq; }
        *p++ = 1u << 4;
        *p++ = 0;
        uint32_t *tok = reinterpret_cast<uint32_t *>(slot + T0);
        uint32_t xbits = 0;
        for (uint32_t k = 0; k < nseq; ++k) {
            uint32_t llc, llb, lle, mlc, mlb, mle;
            rc::zstd_ll_code(seq[k].ll, llc, llb, lle);
            rc::zstd_ml_code(seq[k].ml, mlc, mlb, mle);
            tok[nseq - 1 - k] = llc | (mlc << 6) | (lle << 12) | (mle << 21);
            xbits += llb + mlb;
        }
        raw = (P - 3) + ((xbits + 12 * nseq + 1 + 7) >> 3) >= n;
    }
    if (raw) {
        const uint32_t h = lastbit | (n << 3);
        slot[0] = (uint8_t)h; slot[1] = (uint8_t)(h >> 8); slot[2] = (uint8_t)(h >> 16);
        memcpy(slot + 3, src, n);
        return 3 + n;
    }
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(slot);
    const uint32_t w0 = P >> 2, nb0 = 8 * (P & 3u);
    uint64_t acc = nb0 ? (uint64_t)(slot32[w0] & ((1u << nb0) - 1u)) : 0ull;
    uint32_t o = 0;
    uint32_t nb = rc::fse_chain(reinterpret_cast<const rc::ZW4 *>(slot + T0), nseq, acc, nb0, T,
                                [&](uint32_t v) { slot32[w0 + o] = v; ++o; });  // in place, like the kernel's fallback path
    const uint32_t end = 4 * (w0 + o) + ((nb + 7) >> 3);
    if (nb) { slot32[w0 + o] = (uint32_t)acc; if (nb > 32) slot32[w0 + o + 1] = (uint32_t)(acc >> 32); }
    slot32[0] |= lastbit | (2u << 1) | ((end - 3) << 3);
    return end;
}

extern "C" int64_t zstd_check_encode_frame(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, int streaming)
{
    static rc::ZstdTables T;
    static bool init = false;
    if (!init) { rc::zstd_build_tables(T); init = true; }
    uint8_t *p = dst;
    const uint8_t hdr[6] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x00};  // magic, FHD (no FCS, window descriptor follows), 1 KiB window
    if (cap < 6) return -1;
    memcpy(p, hdr, 6);
    p += 6;
    std::vector<rc::ZstdSeq> seq(rc::ZSTD_BLK / 4 + 2);
    uint8_t tmp[rc::ZSTD_SLOT_MAX + 16];
    if (n == 0) {  // a frame needs at least one block: empty raw block, last
        if ((uint64_t)(p - dst) + 3 > cap) return -1;
        p[0] = 1; p[1] = 0; p[2] = 0;
        return (p - dst) + 3;
    }
    alignas(16) uint8_t slot[640 + 16];
    for (uint64_t o = 0; o < n; o += rc::ZSTD_BLK) {
        const uint32_t len = (uint32_t)((n - o) < rc::ZSTD_BLK ? (n - o) : rc::ZSTD_BLK);
        uint32_t used;
        if (streaming) {  // the form the HIP kernels run
            used = token_form_block(src + o, len, slot, T, o + len >= n);
            memcpy(tmp, slot, used);
        } else
            used = rc::zstd_encode_block(src + o, len, tmp, seq.data(), T, o + len >= n);
        if ((uint64_t)(p - dst) + used > cap) return -1;
        memcpy(p, tmp, used);
        p += used;
    }
    return p - dst;
}

// ---- modelled encoder (rc_zstd_model.h + the serial block forms of rc_zstd_block.h) --------------------------------------
// scalar stand-in for k_zstd_sample: histograms of what the plain parse of `bitmap` (512-byte blocks) and the bytes of `pix` hold
extern "C" uint32_t zm_check_build(const uint8_t *bitmap, uint64_t nb, const uint8_t *pix, uint64_t np, rc::ZstdModel *out)
{
    rc::ZstdSample h;
    memset(&h, 0, sizeof h);
    for (uint64_t o = 0; o < nb; o += rc::ZSTD_BLK) {
        const uint32_t n = (uint32_t)((nb - o) < rc::ZSTD_BLK ? (nb - o) : rc::ZSTD_BLK);
        const uint8_t *src = bitmap + o;
        bool any = false;
        for (uint32_t k = 0; k < n; ++k) any |= src[k] != 0;
        if (any) { for (uint32_t k = 0; k < n; ++k) h.all[src[k]]++; h.nblk++; }   // (what k_zstd_sample counts: rc_zstd.hip)
        uint32_t i = 0, lit_start = 0;
        while (i < n) {
            if (src[i] != 0) { ++i; continue; }
            uint32_t j = i + 1;
            while (j < n && src[j] == 0) ++j;
            if (j - i >= 4) {
                for (uint32_t k = lit_start; k <= i; ++k) h.lit[src[k]]++;
                uint32_t c, b, e;
                rc::zstd_ll_code(i + 1 - lit_start, c, b, e); h.ll[c]++;
                rc::zstd_ml_code(j - i - 1, c, b, e); h.ml[c]++;
                lit_start = j;
            }
            i = j;
        }
        for (uint32_t k = lit_start; k < n; ++k) h.lit[src[k]]++;
    }
    for (uint64_t i = 0; i < np; ++i) h.pix[pix[i]]++;
    rc::zm_build_model(h, *out);
    return out->valid;
}
extern "C" uint64_t zm_model_bytes() { return sizeof(rc::ZstdModel); }

static const uint8_t kFrameHdr[6] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x00};  // 1 KiB window

extern "C" int64_t zm_check_encode_bitmap_frame(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, const rc::ZstdModel *M)
{
    uint8_t *p = dst;
    if (cap < 9) return -1;
    memcpy(p, kFrameHdr, 6);
    p += 6;
    if (n == 0) { p[0] = 1; p[1] = 0; p[2] = 0; return 9; }
    std::vector<rc::ZstdSeq> seq(rc::ZSTD_BLK / 4 + 2);
    uint8_t lits[rc::ZSTD_BLK + 8], hbuf[rc::ZSTD_BLK * 11 / 8 + 16], blk[640 + 16];
    uint32_t have = 0;   // definitions the frame already carries
    for (uint64_t o = 0; o < n; o += rc::ZSTD_BLK) {
        const uint32_t len = (uint32_t)((n - o) < rc::ZSTD_BLK ? (n - o) : rc::ZSTD_BLK);
        uint32_t used = rc::zstd_encode_block_model(src + o, len, blk, seq.data(), lits, hbuf, *M, o + len >= n);
        uint32_t seq_pos;
        const uint32_t add = rc::zm_block_needs(blk, &seq_pos) & ~have;
        have |= add;
        const uint32_t tl = (add & 1u) ? M->lit_desc_len : 0u, sl = (add & 2u) ? M->seq_desc_len : 0u;
        if (used + tl + sl > 640) return -2;   // the budget rule must have prevented this
        if ((uint64_t)(p - dst) + used + tl + sl > cap) return -1;
        for (uint32_t i = 0; i < used + tl + sl; ++i) p[i] = rc::zm_defs_byte(blk, seq_pos, M->lit_desc, tl, M->seq_desc, sl, i);
        p += used + tl + sl;
    }
    return p - dst;
}

extern "C" int64_t zm_check_encode_pix_frame(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, const rc::ZstdModel *M,
                                             uint32_t chunk)
{
    uint8_t *p = dst;
    if (cap < 9 || chunk == 0 || chunk > 1023) return -1;
    memcpy(p, kFrameHdr, 6);
    p += 6;
    if (n == 0) { p[0] = 1; p[1] = 0; p[2] = 0; return 9; }
    std::vector<uint8_t> hbuf(chunk * 11 / 8 + 16), blk(chunk + 32);
    bool have = false;
    for (uint64_t o = 0; o < n; o += chunk) {
        const uint32_t len = (uint32_t)((n - o) < chunk ? (n - o) : chunk);
        const uint32_t used = rc::zm_encode_pix_chunk(src + o, len, blk.data(), hbuf.data(), *M, o + len >= n);
        uint32_t seq_pos;
        const bool add = (rc::zm_block_needs(blk.data(), &seq_pos) & 1u) && !have;
        have |= add;
        const uint32_t tl = add ? M->pix_desc_len : 0u;
        if ((uint64_t)(p - dst) + used + tl > cap) return -1;
        for (uint32_t i = 0; i < used + tl; ++i) p[i] = rc::zm_defs_byte(blk.data(), 0, M->pix_desc, tl, nullptr, 0, i);
        p += used + tl;
    }
    return p - dst;
}
