// rc_zstd_model.h - HOST side of the "modelled" Zstandard encoder: entropy tables fitted to a histogram, in the forms the
// kernels use (code tables) and the forms a zstd frame carries (Huffman tree description, FSE table descriptions).
//
// Why: the reference compresses both per-frame streams with the stock library (pyrecode/recode_writer.py:503-511,
// recode_compressors.py:88), whose blocks carry Huffman-coded literals and FSE tables fitted to the data.  Round 1's device
// encoder used raw literals and the format's predefined FSE tables (ratio 0.24 on the 1 % bitmap against libzstd's 0.143)
// and stored the residual stream.  A zstd frame may define its tables once and reuse them (Treeless_Literals_Block,
// Repeat_Mode: RFC 8878 3.1.1.3.1.1 / 3.1.1.3.2.1), so every 512-byte block can still be encoded independently by one
// wavefront as long as all blocks of a frame agree on the tables BEFOREHAND: a ctx fits one model to a sample of its first
// batch (k_zstd_sample) and every frame it writes carries that model's descriptions in the first block that needs them.
//
// Everything here is plain C++ (no HIP): built into librecode_hip.so for the ctx, and into the CPU format check
// (tests/native/zstd_host_check.cpp) where stock libzstd judges the descriptions.  Written from RFC 8878; the two
// algorithms a decoder forces on an encoder (canonical Huffman code values from weights, FSE state tables from normalised
// counts) follow the specification's decoding tables.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "rc_zstd_block.h"

namespace rc {

// ---- little-endian bit writer ----------------------------------------------------------------------------------------
struct ZmBits {
    uint8_t *p;
    uint32_t cap, n;      // capacity in bytes, bits written
    bool ok;
    ZmBits(uint8_t *dst, uint32_t capacity) : p(dst), cap(capacity), n(0), ok(true) { memset(dst, 0, capacity); }
    void add(uint32_t v, uint32_t nb)
    {
        for (uint32_t i = 0; i < nb; ++i, ++n) {
            if ((n >> 3) >= cap) { ok = false; return; }
            if ((v >> i) & 1u) p[n >> 3] |= (uint8_t)(1u << (n & 7));
        }
    }
    uint32_t bytes() const { return (n + 7) >> 3; }
};

// ---- Huffman ------------------------------------------------------------------------------------------------------------
// Code lengths (1..maxbits) for ALL 256 byte values from a histogram: every value must stay encodable whatever later frames
// contain, so the counts are smoothed (x256 + 1) before the usual pairing; lengths above maxbits are cut and the Kraft sum is
// then repaired so that it is exactly one again (a zstd tree description only exists for complete codes).
inline void zm_huf_lengths(const uint32_t *hist, uint8_t *len, int maxbits = ZM_HUF_MAXBITS)
{
    struct Node { uint64_t w; int l, r; };
    std::vector<Node> nodes;
    std::vector<int> live;
    for (int s = 0; s < 256; ++s) { nodes.push_back({(uint64_t)hist[s] * 256u + 1u, -1, -1}); live.push_back(s); }
    while (live.size() > 1) {   // 256 symbols: the quadratic pairing is cheap enough
        std::sort(live.begin(), live.end(), [&](int a, int b) { return nodes[a].w != nodes[b].w ? nodes[a].w > nodes[b].w : a > b; });
        const int a = live.back(); live.pop_back();
        const int b = live.back(); live.pop_back();
        nodes.push_back({nodes[a].w + nodes[b].w, a, b});
        live.push_back((int)nodes.size() - 1);
    }
    std::vector<int> depth(nodes.size(), 0);
    for (int i = (int)nodes.size() - 1; i >= 256; --i) { depth[nodes[i].l] = depth[i] + 1; depth[nodes[i].r] = depth[i] + 1; }
    int64_t kraft = 0;  // in units of 2^-maxbits
    for (int s = 0; s < 256; ++s) {
        len[s] = (uint8_t)std::min(std::max(depth[s], 1), maxbits);
        kraft += (int64_t)1 << (maxbits - len[s]);
    }
    const int64_t one = (int64_t)1 << maxbits;
    std::vector<int> order(256);
    for (int s = 0; s < 256; ++s) order[s] = s;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hist[a] != hist[b] ? hist[a] < hist[b] : a > b; });  // rarest first
    while (kraft > one) {       // over-subscribed after the cut: lengthen the rarest symbols that still can be
        bool moved = false;
        for (int s : order)
            if (len[s] < maxbits) { kraft -= (int64_t)1 << (maxbits - len[s] - 1); ++len[s]; moved = true; if (kraft <= one) break; }
        if (!moved) break;
    }
    while (kraft < one) {       // slack: shorten the most frequent symbol whose step fits (a maxbits symbol always does)
        bool moved = false;
        for (int i = 255; i >= 0; --i) {
            const int s = order[i];
            const int64_t gain = (int64_t)1 << (maxbits - len[s]);
            if (len[s] > 1 && gain <= one - kraft) { --len[s]; kraft += gain; moved = true; break; }
        }
        if (!moved) break;
    }
}

// Code values the decoder's table construction implies (RFC 8878 4.2.1: symbols in order of increasing weight, then
// increasing value, fill the decoding table from index 0): longest codes get the numerically smallest values.
inline void zm_huf_codes(const uint8_t *len, uint16_t *code_len12, int maxbits = ZM_HUF_MAXBITS)
{
    uint32_t per_len[16] = {0}, next[16] = {0};
    for (int s = 0; s < 256; ++s) per_len[len[s]]++;
    uint32_t min = 0;
    for (int n = maxbits; n >= 1; --n) { next[n] = min; min = (min + per_len[n]) >> 1; }
    for (int s = 0; s < 256; ++s) code_len12[s] = (uint16_t)(next[len[s]]++ | ((uint32_t)len[s] << 12));
}

// ---- FSE (table descriptions and the 2-state stream of the Huffman weights) -------------------------------------------
// Normalised counts summing to 2^log.  `need[s]`: symbol s must stay encodable even if the sample never saw it (probability
// "less than one", -1).  Returns false when the alphabet does not fit the table.
inline bool zm_fse_normalize(const uint32_t *hist, const bool *need, int nsym, int log, int16_t *norm)
{
    const int size = 1 << log;
    uint64_t total = 0;
    for (int s = 0; s < nsym; ++s) total += hist[s];
    int used = 0, big = -1;
    for (int s = 0; s < nsym; ++s) {
        norm[s] = 0;
        if (hist[s] == 0) { if (need[s]) { norm[s] = -1; used += 1; } continue; }
        const uint64_t q = total ? ((uint64_t)hist[s] * size + total / 2) / total : 0;
        if (q <= 1) { norm[s] = -1; used += 1; }   // a one-slot symbol costs `log` bits either way; -1 keeps it out of the spread
        else { norm[s] = (int16_t)q; used += (int)q; }
        if (big < 0 || hist[s] > hist[big]) big = s;
    }
    if (big < 0) {   // empty sample: everything needed gets one slot, the first symbol takes the rest
        for (int s = 0; s < nsym; ++s) if (norm[s] != 0 && big < 0) big = s;
        if (big < 0) return false;
    }
    // give / take the difference to / from the largest symbol(s)
    int diff = size - used;
    if (norm[big] < 0) { norm[big] = 1; }
    while (diff != 0) {
        if (diff > 0) { norm[big] = (int16_t)(norm[big] + diff); diff = 0; break; }
        // too many slots: shrink the largest entries one by one (never below 2 so that they stay in the spread)
        int m = -1;
        for (int s = 0; s < nsym; ++s) if (norm[s] > 2 && (m < 0 || norm[s] > norm[m])) m = s;
        if (m < 0) return false;
        const int take = std::min(-diff, norm[m] - 2);
        norm[m] = (int16_t)(norm[m] - take);
        diff += take;
    }
    return true;
}

// FSE table description (RFC 8878 4.1.1): 4 bits Accuracy_Log - 5, then each symbol's count + 1 in a field whose width
// shrinks with the remaining probability mass; a zero count is followed by 2-bit repeat flags.
inline bool zm_fse_write_ncount(const int16_t *norm, int nsym, int log, uint8_t *dst, uint32_t cap, uint32_t *len)
{
    ZmBits b(dst, cap);
    b.add((uint32_t)(log - 5), 4);
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1;
    int last = nsym - 1;
    while (last > 0 && norm[last] == 0) --last;
    int s = 0;
    bool prev0 = false;
    while (s <= last && remaining > 1) {
        if (prev0) {
            int start = s;
            while (s <= last && norm[s] == 0) ++s;
            if (s > last) return false;
            while (s >= start + 3) { b.add(3, 2); start += 3; }
            b.add((uint32_t)(s - start), 2);
        }
        int count = norm[s++];
        const int max = (2 * threshold - 1) - remaining;
        remaining -= count < 0 ? -count : count;
        ++count;
        if (count >= threshold) count += max;
        b.add((uint32_t)count, (uint32_t)(nbits - (count < max ? 1 : 0)));
        prev0 = count == 1;
        if (remaining < 1) return false;
        while (remaining < threshold) { --nbits; threshold >>= 1; }
    }
    if (remaining != 1 || !b.ok) return false;
    *len = b.bytes();
    return true;
}

// compression table of a normalised distribution (any Accuracy_Log <= 9): the spread and state numbering every decoder
// derives from the same counts (RFC 8878 4.1.1 "from normalised distribution to decoding tables")
inline void zm_fse_ctable(const int16_t *norm, int nsym, int log, uint16_t *state_table, uint32_t *dnb, int32_t *dfs)
{
    const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    std::vector<uint8_t> symbol(size);
    std::vector<int> cumul(nsym + 2);
    int high = size - 1;
    cumul[0] = 0;
    for (int u = 1; u <= nsym; ++u) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; symbol[high--] = (uint8_t)(u - 1); }
        else cumul[u] = cumul[u - 1] + norm[u - 1];
    }
    int pos = 0;
    for (int s = 0; s < nsym; ++s)
        for (int i = 0; i < norm[s]; ++i) {
            symbol[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    {
        std::vector<int> c2(cumul);
        for (int u = 0; u < size; ++u) state_table[c2[symbol[u]]++] = (uint16_t)(size + u);
    }
    int total = 0;
    for (int s = 0; s < nsym; ++s) {
        const int n = norm[s];
        if (n == 0) { dnb[s] = (uint32_t)(((log + 1) << 16) - (1 << log)); dfs[s] = 0; }
        else if (n == -1 || n == 1) { dnb[s] = (uint32_t)((log << 16) - (1 << log)); dfs[s] = total - 1; total++; }
        else {
            const int hb = 31 - __builtin_clz((unsigned)(n - 1));
            const int max_bits = log - hb;
            dnb[s] = (uint32_t)((max_bits << 16) - (n << max_bits));
            dfs[s] = total - n;
            total += n;
        }
    }
}

// Huffman tree description (RFC 8878 4.2.1): the weights of all symbols but the last, FSE-compressed with two interleaved
// states (headerByte < 128 = the compressed size).  weight = (longest code length) + 1 - length.  Returns false if it does not fit the
// format's 127-byte limit (then the caller does without Huffman-coded literals).
inline bool zm_huf_write_desc(const uint8_t *len, uint8_t *dst, uint32_t cap, uint32_t *out_len)
{
    int maxbits = 1;   // the tree's own depth: a decoder insists on (an even number of) weight-1 symbols
    for (int s = 0; s < 256; ++s) maxbits = std::max(maxbits, (int)len[s]);
    uint8_t w[255];
    uint32_t hist[16] = {0};
    bool need[16] = {false};
    for (int s = 0; s < 255; ++s) { w[s] = (uint8_t)(maxbits + 1 - len[s]); hist[w[s]]++; }
    int nsym = 13;
    while (nsym > 1 && hist[nsym - 1] == 0) --nsym;
    const int log = 6;   // the format's maximum for the weights
    int16_t norm[16];
    if (!zm_fse_normalize(hist, need, nsym, log, norm)) return false;
    uint8_t tmp[ZM_DESC_MAX];
    uint32_t nc_len = 0;
    if (!zm_fse_write_ncount(norm, nsym, log, tmp, sizeof tmp, &nc_len)) return false;
    uint16_t st[64];
    uint32_t dnb[16];
    int32_t dfs[16];
    zm_fse_ctable(norm, nsym, log, st, dnb, dfs);
    // two states, the source walked from its end; the LAST state flushed is the FIRST the decoder reads (state 1, which
    // decodes weight 0): with an odd count state 1 starts one symbol ahead so that the alternation ends on weight 0
    uint8_t bits[ZM_DESC_MAX];
    ZmBits b(bits, sizeof bits);
    auto init = [&](uint32_t sym) {
        const uint32_t nb = (dnb[sym] + (1u << 15)) >> 16;
        const uint32_t v = (nb << 16) - dnb[sym];
        return (uint32_t)st[(int32_t)(v >> nb) + dfs[sym]];
    };
    auto enc = [&](uint32_t &state, uint32_t sym) {
        const uint32_t nb = (state + dnb[sym]) >> 16;
        b.add(state & ((1u << nb) - 1u), nb);
        state = st[(int32_t)(state >> nb) + dfs[sym]];
    };
    int ip = 255;
    uint32_t s1, s2;
    {   // 255 weights: odd
        s1 = init(w[--ip]);
        s2 = init(w[--ip]);
        enc(s1, w[--ip]);
    }
    while (ip > 0) {
        enc(s2, w[--ip]);
        if (ip == 0) break;   // (cannot happen for an even remainder; kept for clarity)
        enc(s1, w[--ip]);
    }
    b.add(s2 & 63u, 6);
    b.add(s1 & 63u, 6);
    b.add(1, 1);
    if (!b.ok) return false;
    const uint32_t total = nc_len + b.bytes();
    if (total >= 128 || 1 + total > cap) return false;
    dst[0] = (uint8_t)total;
    memcpy(dst + 1, tmp, nc_len);
    memcpy(dst + 1 + nc_len, bits, b.bytes());
    *out_len = 1 + total;
    return true;
}

// ---- the model ----------------------------------------------------------------------------------------------------------
// Extra bits of the literal-length / match-length codes (RFC 8878 3.1.1.3.2.1.1), for the size estimate below.
inline uint32_t zm_ll_extra(int c) { static const uint8_t b[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16}; return c < 36 ? b[c] : 16; }
inline uint32_t zm_ml_extra(int c) { static const uint8_t b[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16}; return c < 53 ? b[c] : 16; }
// bits the sample's sequences cost under normalised counts `norm` (table log `log`): state bits ~ log - log2(count) per symbol
inline double zm_fse_cost(const uint32_t *hist, const int16_t *norm, int nsym, int log)
{
    double bits = 0;
    for (int s = 0; s < nsym; ++s) {
        if (!hist[s]) continue;
        const double cnt = norm[s] > 0 ? (double)norm[s] : 1.0;
        double l2 = 0;
        for (double v = (double)(1 << log) / cnt; v > 1.0; v /= 2.0) l2 += v >= 2.0 ? 1.0 : (v - 1.0);   // ~log2, no libm
        bits += hist[s] * l2;
    }
    return bits;
}

inline void zm_build_model(const ZstdSample &h, ZstdModel &m, uint32_t speed_permille = 0)
{
    memset(&m, 0, sizeof m);
    uint8_t len[256];
    zm_huf_lengths(h.lit, len);
    // Two block forms for the binary maps (rc_zstd_wave.h): literals + sequences (zero runs as matches: what pays on sparse maps) or ALL
    // bytes as Huffman-coded literals and no sequences.  On dense maps (a few per cent of the pixels set: most runs are short, a third
    // of the bytes are literals anyway) the second form is SMALLER - a byte-wise code reaches the maps' entropy to within a tenth where
    // the sequences cost 10-20 bits each - and needs neither the FSE chain kernel nor the sequence tables.  Decided once per model,
    // from the sample: estimated bits of both forms.  speed_permille: how much LARGER (in 1/1000 of the sequences form's size) the
    // literals-only form may be and still be chosen - what a low compression_level is for (the caller maps the level: rc_api.hip):
    // the form without sequences saves the whole FSE pass and a third of the tokenizer, +13 % frames/s at 4096^2.
    double bits_seq = 0, bits_all = 0;
    {
        uint8_t len_all[256];
        zm_huf_lengths(h.all, len_all);
        for (int v = 0; v < 256; ++v) { bits_seq += (double)h.lit[v] * len[v]; bits_all += (double)h.all[v] * len_all[v]; }
        bits_seq += 8.0 * 9 * h.nblk;    // block header 3, literals header 3, count 1, modes 1, ~1 byte of initial states and padding
        bits_all += 8.0 * 7 * h.nblk;    // block header 3, literals header 3, count 1
        bool need_ll[ZM_LL_SYMS], need_ml[ZM_ML_SYMS];
        for (int s = 0; s < ZM_LL_SYMS; ++s) need_ll[s] = s <= 28;
        for (int s = 0; s < ZM_ML_SYMS; ++s) need_ml[s] = s <= 45;
        int16_t nll[ZM_LL_SYMS], nml[ZM_ML_SYMS];
        if (zm_fse_normalize(h.ll, need_ll, ZM_LL_SYMS, 8, nll) && zm_fse_normalize(h.ml, need_ml, ZM_ML_SYMS, 9, nml))
            bits_seq += zm_fse_cost(h.ll, nll, ZM_LL_SYMS, 8) + zm_fse_cost(h.ml, nml, ZM_ML_SYMS, 9);
        for (int c = 0; c < ZM_LL_SYMS; ++c) bits_seq += (double)h.ll[c] * zm_ll_extra(c);
        for (int c = 0; c < ZM_ML_SYMS; ++c) bits_seq += (double)h.ml[c] * zm_ml_extra(c);
        if (h.nblk && bits_all * 1000.0 < bits_seq * (1000.0 + speed_permille)) {
            m.valid |= ZM_LITS_ONLY;
            memcpy(len, len_all, sizeof len);
        }
    }
    zm_huf_codes(len, m.lit_code);
    if (zm_huf_write_desc(len, m.lit_desc, sizeof m.lit_desc, &m.lit_desc_len)) m.valid |= 1u;
    else m.valid &= ~ZM_LITS_ONLY;
    zm_huf_lengths(h.pix, len);
    zm_huf_codes(len, m.pix_code);
    if (zm_huf_write_desc(len, m.pix_desc, sizeof m.pix_desc, &m.pix_desc_len)) m.valid |= 2u;
    // literal-length codes 0..27 and match-length codes 0..44 are what a 512-byte block can produce; all of them stay
    // encodable (one slot each at least)
    bool need_ll[ZM_LL_SYMS], need_ml[ZM_ML_SYMS];
    for (int s = 0; s < ZM_LL_SYMS; ++s) need_ll[s] = s <= 28;
    for (int s = 0; s < ZM_ML_SYMS; ++s) need_ml[s] = s <= 45;
    int16_t nll[ZM_LL_SYMS], nml[ZM_ML_SYMS];
    const int ll_log = 8, ml_log = 9;
    uint32_t l1 = 0, l2 = 0;
    if (zm_fse_normalize(h.ll, need_ll, ZM_LL_SYMS, ll_log, nll) && zm_fse_normalize(h.ml, need_ml, ZM_ML_SYMS, ml_log, nml) &&
        zm_fse_write_ncount(nll, ZM_LL_SYMS, ll_log, m.seq_desc, ZM_DESC_MAX - 2, &l1) &&
        zm_fse_write_ncount(nml, ZM_ML_SYMS, ml_log, m.seq_desc + l1 + 1, ZM_DESC_MAX - l1 - 1, &l2)) {
        m.seq_desc[l1] = 0;   // Offsets: RLE mode, the one code is 0 ("repeat offset 1", no extra bits)
        m.seq_desc_len = l1 + 1 + l2;
        zm_fse_ctable(nll, ZM_LL_SYMS, ll_log, m.seq.ll_state, m.seq.ll_dnb, m.seq.ll_dfs);
        zm_fse_ctable(nml, ZM_ML_SYMS, ml_log, m.seq.ml_state, m.seq.ml_dnb, m.seq.ml_dfs);
        m.seq.ll_log = ll_log;
        m.seq.ml_log = ml_log;
        if (!(m.valid & ZM_LITS_ONLY)) m.valid |= 4u;   // (literals only: no block carries sequences, and none has to leave room for their tables)
    }
}

}  // namespace rc
