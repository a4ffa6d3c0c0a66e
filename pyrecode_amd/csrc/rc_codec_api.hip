// rc_codec_api.hip - the stateless codec seam (include/recode_hip.h, seam 2): rc_compress / rc_decompress / rc_compress_bound for
// LZ4 frames, zstd frames and blosc1-LZ4 chunks of arbitrary buffers, on the caller's current GPU (utility context, rc_host.h).
// Replaces compress() / de_compress() of pyrecode/recode_compressors.py:82-120, 40-79 for the schemes with a device codec.
#include "rc_host.h"

// ---- seam 2 ----------------------------------------------------------------------------------------------------
static int lz4_compress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n, uint32_t level)
{
    using namespace rc;
    Util &u = g_util;
    if (n >= (1ull << 32)) return fail(RC_ERR_BAD_ARG, "rc_compress: input must be < 4 GiB");
    if (n == 0) {  // empty frame: header + EndMark
        const uint32_t h = lz4f_descriptor(0x40);
        const uint8_t f[11] = {0x04, 0x22, 0x4D, 0x18, (uint8_t)h, (uint8_t)(h >> 8), (uint8_t)(h >> 16), 0, 0, 0, 0};
        if (dst_cap < 11) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small");
        HIP_TRY(hipMemcpy(dst, f, 11, is_device_ptr(dst) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        *out_n = 11;
        return RC_OK;
    }
    Scratch sc;
    sc.ntiles = (uint32_t)((n + TILE_BM - 1) / TILE_BM);
    sc.nb = n;
    sc.nb_stride = (uint64_t)sc.ntiles * TILE_BM;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src, sc.nb_stride - n + 16);
    if (r != RC_OK) return r;
    sc.bitmap = const_cast<uint8_t *>(d_src);
    const uint64_t T = sc.ntiles;
    r = ensure(u.w, u.w_cap, T * BLK_SLOT + T * 8 + 64);
    if (r != RC_OK) return r;
    sc.blk_slots = u.w;
    sc.blk_size = reinterpret_cast<uint32_t *>(u.w + T * BLK_SLOT);
    sc.blk_off = sc.blk_size + T;
    sc.frame_cbytes = sc.blk_off + T;
    launch_lz4_encode_buffer(sc, u.stream, level != 0);   // level 0: zero runs only; >= 1: the event parser (rc_lz4_block.h)
    launch_scans(sc, 1, false, true, u.stream);
    HIP_TRY(hipMemcpyAsync(u.h_scalar, sc.frame_cbytes, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    const uint64_t total = 7ull + *reinterpret_cast<uint32_t *>(u.h_scalar) + 4;
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small (see rc_compress_bound)");
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_lz4f_gather(sc, lz4f_descriptor(0x40), d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    *out_n = total;
    return RC_OK;
}

static int lz4_decompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    // the frame and block headers are walked on the host (sequential by format, a few bytes per block)
    std::vector<uint8_t> hsrc;
    const uint8_t *h = src;
    if (is_device_ptr(src)) {
        hsrc.resize(n);
        HIP_TRY(hipMemcpy(hsrc.data(), src, n, hipMemcpyDeviceToHost));
        h = hsrc.data();
    }
    auto rd32 = [&](uint64_t p) { return (uint32_t)h[p] | ((uint32_t)h[p + 1] << 8) | ((uint32_t)h[p + 2] << 16) | ((uint32_t)h[p + 3] << 24); };
    if (n < 11 || rd32(0) != 0x184D2204u) return fail(RC_ERR_CORRUPT, "not an LZ4 frame");
    const uint32_t flg = h[4], bd = h[5];
    if ((flg >> 6) != 1 || (flg & 2) || (bd & 0x8F) || ((bd >> 4) & 7) < 4) return fail(RC_ERR_CORRUPT, "bad LZ4 frame descriptor");
    const int linked = !((flg >> 5) & 1), bsum = (flg >> 4) & 1, csize = (flg >> 3) & 1, csum = (flg >> 2) & 1, dict = flg & 1;
    const uint64_t bmax = 1ull << (8 + 2 * ((bd >> 4) & 7));
    uint64_t ip = 6 + (csize ? 8 : 0) + (dict ? 4 : 0) + 1;
    std::vector<Lz4Block> blks;
    for (;;) {
        if (ip + 4 > n) return fail(RC_ERR_CORRUPT, "truncated LZ4 frame");
        uint32_t bs = rd32(ip);
        ip += 4;
        if (bs == 0) break;
        const uint32_t raw = bs >> 31;
        bs &= 0x7FFFFFFFu;
        if (bs > bmax || ip + bs > n) return fail(RC_ERR_CORRUPT, "LZ4 block exceeds the frame");
        blks.push_back(Lz4Block{ip, bs, raw});
        ip += bs + (bsum ? 4 : 0);
    }
    if (csum) ip += 4;
    if (ip != n) return fail(RC_ERR_CORRUPT, "trailing bytes after the LZ4 frame");
    const uint32_t nblk = (uint32_t)blks.size();
    if (nblk == 0) { *out_n = 0; return RC_OK; }
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src);
    if (r != RC_OK) return r;
    // work buffer: block table | sizes | offsets | err
    const uint64_t tab = (uint64_t)nblk * sizeof(Lz4Block), szs = ((uint64_t)nblk * 4 + 7) & ~7ull, offs = (uint64_t)nblk * 8;
    r = ensure(u.w, u.w_cap, tab + szs + offs + 16);
    if (r != RC_OK) return r;
    Lz4Block *d_blks = reinterpret_cast<Lz4Block *>(u.w);
    uint32_t *d_sizes = reinterpret_cast<uint32_t *>(u.w + tab);
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(u.w + tab + szs);
    int *d_err = reinterpret_cast<int *>(u.w + tab + szs + offs);
    HIP_TRY(hipMemcpyAsync(d_blks, blks.data(), tab, hipMemcpyHostToDevice, u.stream));
    HIP_TRY(hipMemsetAsync(d_err, 0, 4, u.stream));
    launch_lz4_decode(d_src, d_blks, nblk, d_sizes, nullptr, nullptr, ~0ull, linked, d_err, u.stream);
    std::vector<uint32_t> sizes(nblk);
    int err = 0;
    HIP_TRY(hipMemcpyAsync(sizes.data(), d_sizes, (uint64_t)(linked ? 1 : nblk) * 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block");
    std::vector<uint64_t> off(nblk, 0);
    uint64_t total = 0;
    if (linked) total = sizes[0];
    else
        for (uint32_t b = 0; b < nblk; ++b) {
            if (sizes[b] > bmax) return fail(RC_ERR_CORRUPT, "LZ4 block decodes beyond its declared maximum");
            off[b] = total;
            total += sizes[b];
        }
    *out_n = total;  // reported even when dst is too small, so a caller can size its buffer and call again
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small");
    if (total == 0) return RC_OK;
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    HIP_TRY(hipMemcpyAsync(d_offs, off.data(), offs, hipMemcpyHostToDevice, u.stream));
    uint32_t max_stored = 0;
    for (const Lz4Block &q : blks) if (q.raw) max_stored = std::max(max_stored, q.size);
    launch_lz4_decode(d_src, d_blks, nblk, nullptr, d_offs, d_out, total, linked, d_err, u.stream, max_stored);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block");
    return RC_OK;
}

static int zstd_compress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    if (n >= (1ull << 32)) return fail(RC_ERR_BAD_ARG, "rc_compress: input must be < 4 GiB");
    if (n == 0) {  // a frame needs one block: empty raw block with Last_Block
        const uint8_t f[9] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x00, 0x01, 0x00, 0x00};
        if (dst_cap < 9) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small");
        HIP_TRY(hipMemcpy(dst, f, 9, is_device_ptr(dst) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        *out_n = 9;
        return RC_OK;
    }
    Scratch sc;
    sc.ntiles = (uint32_t)((n + TILE_BM - 1) / TILE_BM);
    sc.nb = n;
    sc.nb_stride = (uint64_t)sc.ntiles * TILE_BM;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src, sc.nb_stride - n + 16);
    if (r != RC_OK) return r;
    sc.bitmap = const_cast<uint8_t *>(d_src);
    const uint64_t T = sc.ntiles;
    r = ensure(u.w, u.w_cap, T * BLK_SLOT + T * 8 + 64);
    if (r != RC_OK) return r;
    sc.blk_slots = u.w;
    sc.blk_size = reinterpret_cast<uint32_t *>(u.w + T * BLK_SLOT);
    sc.blk_off = sc.blk_size + T;
    sc.frame_cbytes = sc.blk_off + T;
    if (!u.ztab) {
        std::vector<uint8_t> tab(zstd_tables_bytes());
        zstd_tables_host(tab.data());
        HIP_TRY(hipMalloc(&u.ztab, tab.size()));
        HIP_TRY(hipMemcpy(u.ztab, tab.data(), tab.size(), hipMemcpyHostToDevice));
    }
    launch_zstd_encode_blocks(sc, 1, u.ztab, u.stream);
    launch_scans(sc, 1, false, true, u.stream);
    HIP_TRY(hipMemcpyAsync(u.h_scalar, sc.frame_cbytes, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    const uint64_t total = 6ull + *reinterpret_cast<uint32_t *>(u.h_scalar);
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small (see rc_compress_bound)");
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_zstd_gather(sc, d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    *out_n = total;
    return RC_OK;
}

static int blosc_compress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    if (n >= (1ull << 31) - 16) return fail(RC_ERR_BAD_ARG, "rc_compress: a blosc1 chunk holds < 2 GiB");
    if (n == 0) {  // header only, "memcpyed"
        const uint8_t f[16] = {2, 1, 0x36, 8, 0, 0, 0, 0, 0, 0, 0, 0, 16, 0, 0, 0};
        if (dst_cap < 16) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small");
        HIP_TRY(hipMemcpy(dst, f, 16, is_device_ptr(dst) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        *out_n = 16;
        return RC_OK;
    }
    Scratch sc;
    sc.ntiles = (uint32_t)((n + TILE_BM - 1) / TILE_BM);
    sc.nb = n;
    sc.nb_stride = (uint64_t)sc.ntiles * TILE_BM;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src, sc.nb_stride - n + 16);
    if (r != RC_OK) return r;
    sc.bitmap = const_cast<uint8_t *>(d_src);
    const uint64_t T = sc.ntiles;
    r = ensure(u.w, u.w_cap, T * BLK_SLOT + T * 8 + 64);
    if (r != RC_OK) return r;
    sc.blk_slots = u.w;
    sc.blk_size = reinterpret_cast<uint32_t *>(u.w + T * BLK_SLOT);
    sc.blk_off = sc.blk_size + T;
    sc.frame_cbytes = sc.blk_off + T;
    launch_blosc_encode_blocks(sc, 1, u.stream);
    launch_scans(sc, 1, false, true, u.stream);
    HIP_TRY(hipMemcpyAsync(u.h_scalar, sc.frame_cbytes, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    const uint64_t total = 16ull + 4ull * T + *reinterpret_cast<uint32_t *>(u.h_scalar);
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_compress: dst too small (see rc_compress_bound)");
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, total);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_blosc_gather(sc, d_out, u.stream);
    HIP_TRY(hipGetLastError());
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, total, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    *out_n = total;
    return RC_OK;
}

RC_EXPORT int rc_compress(uint32_t scheme, uint32_t level, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap,
                          uint64_t *out_n)
{
    // level: LZ4 0 = the run parser, >= 1 = the event parser; zstd / blosc through this stateless seam: one effort (the ctx's zstd
    // encoder has the modelled form for level >= 1)
    if (!dst || !out_n || (!src && n)) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (scheme == RC_SCHEME_BLOSC_LZ4) {
        UtilScope util_scope;
        int r = util_scope.enter();
        if (r != RC_OK) return r;
        return blosc_compress(src, n, dst, dst_cap, out_n);
    }
    if (scheme != RC_SCHEME_LZ4 && scheme != RC_SCHEME_ZSTD)
        return fail(RC_ERR_UNSUPPORTED, "rc_compress: compression scheme not implemented on device");
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    return scheme == RC_SCHEME_LZ4 ? lz4_compress(src, n, dst, dst_cap, out_n, level) : zstd_compress(src, n, dst, dst_cap, out_n);
}
// blosc1 chunk with the LZ4 codec (what rc_compress(8) and python-blosc's cname='lz4' write): header and block table are
// walked on the host, the LZ4 blocks are decoded on the GPU into an image of the shuffled chunk, a second kernel unshuffles.
static int blosc_decompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    std::vector<uint8_t> hsrc;
    const uint8_t *h = src;
    if (is_device_ptr(src)) {
        hsrc.resize(n);
        HIP_TRY(hipMemcpy(hsrc.data(), src, n, hipMemcpyDeviceToHost));
        h = hsrc.data();
    }
    auto rd32 = [&](uint64_t p) { return (uint32_t)h[p] | ((uint32_t)h[p + 1] << 8) | ((uint32_t)h[p + 2] << 16) | ((uint32_t)h[p + 3] << 24); };
    if (n < 16 || h[0] != 2) return fail(RC_ERR_CORRUPT, "not a blosc1 chunk");
    const uint32_t flags = h[2], typesize = h[3] ? h[3] : 1;
    const uint64_t nbytes = rd32(4), blocksize = rd32(8), cbytes = rd32(12);
    if (cbytes != n || nbytes >= (1ull << 31)) return fail(RC_ERR_CORRUPT, "blosc1 header disagrees with the chunk length");
    *out_n = nbytes;
    if (nbytes > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small");
    if (nbytes == 0) return RC_OK;
    if (flags & 0x02) {  // memcpyed
        if (n != 16 + nbytes) return fail(RC_ERR_CORRUPT, "bad memcpyed blosc1 chunk");
        HIP_TRY(hipMemcpy(dst, src + 16, nbytes, hipMemcpyDefault));
        return RC_OK;
    }
    if ((flags >> 5) != 1) return fail(RC_ERR_UNSUPPORTED, "blosc1 chunk: only the LZ4 codec is decoded on device");
    if (blocksize == 0 || blocksize > nbytes) return fail(RC_ERR_CORRUPT, "bad blosc1 blocksize");
    const uint64_t nblocks = (nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * nblocks > n) return fail(RC_ERR_CORRUPT, "truncated blosc1 chunk");
    std::vector<Lz4Block> blks;
    std::vector<uint64_t> offs;
    std::vector<uint32_t> want;
    for (uint64_t b = 0; b < nblocks; ++b) {
        const uint64_t bsize = std::min<uint64_t>(blocksize, nbytes - b * blocksize);
        const bool leftover = bsize != blocksize;
        const bool split = !(flags & 0x10) && typesize <= 16 && blocksize / typesize >= 128 && !leftover;  // blosc.c blosc_d
        const uint32_t nsplits = split ? typesize : 1;
        const uint64_t neblock = bsize / nsplits;
        uint64_t pos = rd32(16 + 4 * b);
        for (uint32_t j = 0; j < nsplits; ++j) {
            if (pos + 4 > n) return fail(RC_ERR_CORRUPT, "blosc1 block table points outside the chunk");
            const uint32_t cs = rd32(pos);
            pos += 4;
            if (pos + cs > n || cs > neblock + neblock / 255 + 16) return fail(RC_ERR_CORRUPT, "blosc1 block exceeds the chunk");
            blks.push_back(Lz4Block{pos, cs, cs == neblock ? 1u : 0u});
            offs.push_back(b * blocksize + j * neblock);
            want.push_back((uint32_t)neblock);
            pos += cs;
        }
    }
    const uint32_t nb = (uint32_t)blks.size();
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src);
    if (r != RC_OK) return r;
    const uint64_t tab = (uint64_t)nb * sizeof(Lz4Block), szs = ((uint64_t)nb * 4 + 7) & ~7ull, ofs = (uint64_t)nb * 8;
    r = ensure(u.w, u.w_cap, tab + szs + ofs + 16);
    if (r != RC_OK) return r;
    r = ensure(u.b, u.b_cap, nbytes + 16);  // image of the shuffled chunk
    if (r != RC_OK) return r;
    Lz4Block *d_blks = reinterpret_cast<Lz4Block *>(u.w);
    uint32_t *d_sizes = reinterpret_cast<uint32_t *>(u.w + tab);
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(u.w + tab + szs);
    int *d_err = reinterpret_cast<int *>(u.w + tab + szs + ofs);
    HIP_TRY(hipMemcpyAsync(d_blks, blks.data(), tab, hipMemcpyHostToDevice, u.stream));
    HIP_TRY(hipMemcpyAsync(d_offs, offs.data(), ofs, hipMemcpyHostToDevice, u.stream));
    HIP_TRY(hipMemsetAsync(d_err, 0, 4, u.stream));
    launch_lz4_decode(d_src, d_blks, nb, d_sizes, nullptr, nullptr, ~0ull, 0, d_err, u.stream);  // sizes only: must equal the split size
    std::vector<uint32_t> sizes(nb);
    int err = 0;
    HIP_TRY(hipMemcpyAsync(sizes.data(), d_sizes, (uint64_t)nb * 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block inside the blosc1 chunk");
    for (uint32_t i = 0; i < nb; ++i)
        if (sizes[i] != want[i]) return fail(RC_ERR_CORRUPT, "blosc1 block decodes to the wrong size");
    uint32_t max_stored = 0;
    for (const Lz4Block &q : blks) if (q.raw) max_stored = std::max(max_stored, q.size);
    launch_lz4_decode(d_src, d_blks, nb, nullptr, d_offs, u.b, nbytes, 0, d_err, u.stream, max_stored);
    uint8_t *d_out = dst;
    const bool out_host = !is_device_ptr(dst);
    if (out_host) {
        r = ensure(u.o, u.o_cap, nbytes);
        if (r != RC_OK) return r;
        d_out = u.o;
    }
    launch_blosc_unshuffle(u.b, d_out, nbytes, (uint32_t)blocksize, typesize, (flags & 0x04) ? 4u : ((flags & 0x01) ? 1u : 0u), u.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, u.stream));
    if (out_host) HIP_TRY(hipMemcpyAsync(dst, d_out, nbytes, hipMemcpyDeviceToHost, u.stream));
    HIP_TRY(hipStreamSynchronize(u.stream));
    if (err) return fail(RC_ERR_CORRUPT, "malformed LZ4 block inside the blosc1 chunk");
    return RC_OK;
}

// zstd frame of the subset the device decoders cover (rc_zstd_dec.h: everything rc_compress / the ctx write): the host walks
// the block headers and builds the tables, one lane decodes one block.  The decoded size is not in the frame: the last block
// is decoded "up to" a block's size and reports what it produced.
static int zstd_decompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    using namespace rc;
    Util &u = g_util;
    std::vector<uint8_t> hsrc;
    const uint8_t *h = src;
    if (is_device_ptr(src)) {
        hsrc.resize(n);
        HIP_TRY(hipMemcpy(hsrc.data(), src, n, hipMemcpyDeviceToHost));
        h = hsrc.data();
    }
    std::vector<ZdBlock> all, comp, raw;
    ZdTables T;
    uint64_t bound = 0;
    const int zr = zd_index_frame(h, 0, n, 0, TILE_BM, ~0ull, all, T, &bound);
    if (zr == ZD_FOREIGN) return fail(RC_ERR_UNSUPPORTED, "rc_decompress: zstd stream outside the device decoder's subset (use the stock decoder)");
    if (zr != ZD_OK) return fail(RC_ERR_CORRUPT, "malformed zstd frame");
    uint32_t raw_max = 0;
    for (const ZdBlock &b : all) {
        if (b.type == 2) { if (b.regen > 1024) return fail(RC_ERR_UNSUPPORTED, "rc_decompress: zstd block larger than the device decoder's rows"); comp.push_back(b); }
        else { raw.push_back(b); raw_max = std::max(raw_max, b.regen); }
    }
    // nothing is allocated or zeroed beyond what the caller's buffer justifies: a few KB of RLE blocks can announce gigabytes
    if (bound > dst_cap + 1024) { *out_n = bound; return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small (out_n = an upper bound of the decoded size)"); }
    uint32_t row = TILE_BM;
    for (const ZdBlock &b : comp) if (b.regen > (uint32_t)TILE_BM) row = 1024;
    const uint8_t *d_src = nullptr;
    int r = stage_in(src, n, u.a, u.a_cap, d_src);
    if (r != RC_OK) return r;
    const uint64_t sz_blk = (comp.size() + raw.size()) * sizeof(ZdBlock) + 64;
    if ((r = ensure(u.x[2], u.x_cap[2], sz_blk)) != RC_OK || (r = ensure(u.x[3], u.x_cap[3], sizeof(ZdTables) + 64)) != RC_OK ||
        (r = ensure(u.x[4], u.x_cap[4], 256)) != RC_OK || (r = ensure(u.x[1], u.x_cap[1], bound + 64)) != RC_OK)
        return r;
    if (!u.zd_predef) {
        std::vector<uint8_t> t(zd_tables_bytes());
        zd_predefined_tables(t.data());
        HIP_TRY(hipMalloc(&u.zd_predef, t.size()));
        HIP_TRY(hipMemcpy(u.zd_predef, t.data(), t.size(), hipMemcpyHostToDevice));
    }
    hipStream_t s = u.stream;
    ZdBlock *d_comp = reinterpret_cast<ZdBlock *>(u.x[2]), *d_raw = d_comp + comp.size();
    ZdFrameList *d_lists = reinterpret_cast<ZdFrameList *>(u.x[4]);   // [0] compressed, [1] stored / RLE
    uint64_t *d_base = reinterpret_cast<uint64_t *>(u.x[4] + 32);
    int *d_err = reinterpret_cast<int *>(u.x[4] + 48);
    uint32_t *d_prod = reinterpret_cast<uint32_t *>(u.x[4] + 56);
    const ZdFrameList lists[2] = {{d_comp, (uint32_t)comp.size(), 0}, {d_raw, (uint32_t)raw.size(), 0}};
    const uint64_t base0 = 0;
    const uint32_t none = 0xFFFFFFFFu;
    if (!comp.empty()) HIP_TRY(hipMemcpyAsync(d_comp, comp.data(), comp.size() * sizeof(ZdBlock), hipMemcpyHostToDevice, s));
    if (!raw.empty()) HIP_TRY(hipMemcpyAsync(d_raw, raw.data(), raw.size() * sizeof(ZdBlock), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(u.x[3], &T, sizeof T, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_lists, lists, sizeof lists, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_base, &base0, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(d_err, 0, 4, s));
    HIP_TRY(hipMemsetAsync(u.x[1], 0, bound + 64, s));   // the decoders store only what is not zero
    HIP_TRY(hipMemcpyAsync(d_prod, &none, 4, hipMemcpyHostToDevice, s));
    if (!comp.empty()) launch_block_decode(1, (int)row, d_src, d_lists, 1, (uint32_t)comp.size(), u.x[3], u.zd_predef, u.x[1], d_base, d_err, s, d_prod);
    launch_block_copy(d_src, d_lists + 1, 1, (uint32_t)raw.size(), raw_max, u.x[1], d_base, s);
    HIP_TRY(hipGetLastError());
    int err = 0;
    uint32_t prod = none;
    HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&prod, d_prod, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (err) return fail(RC_ERR_CORRUPT, "malformed zstd block");
    uint64_t total = bound;
    if (prod != none) {   // the flexible last block produced `prod` of the `regen` bytes it was given
        for (const ZdBlock &b : comp) if (b.flex) total = bound - (b.regen - prod);
    }
    *out_n = total;
    if (total > dst_cap) return fail(RC_ERR_OUT_TOO_SMALL, "rc_decompress: dst too small");
    if (total) HIP_TRY(hipMemcpy(dst, u.x[1], total, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return RC_OK;
}

RC_EXPORT int rc_decompress(uint32_t scheme, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n)
{
    if (!src || !out_n || (!dst && dst_cap)) return fail(RC_ERR_BAD_ARG, "NULL argument");
    if (scheme != RC_SCHEME_LZ4 && scheme != RC_SCHEME_BLOSC_LZ4 && scheme != RC_SCHEME_ZSTD)
        return fail(RC_ERR_UNSUPPORTED, "rc_decompress: compression scheme not implemented on device");
    UtilScope util_scope;
    int r = util_scope.enter();
    if (r != RC_OK) return r;
    if (scheme == RC_SCHEME_ZSTD) return zstd_decompress(src, n, dst, dst_cap, out_n);   // RC_ERR_UNSUPPORTED for foreign frames: the
    return scheme == RC_SCHEME_LZ4 ? lz4_decompress(src, n, dst, dst_cap, out_n) : blosc_decompress(src, n, dst, dst_cap, out_n);   // caller's stock decoder
}
RC_EXPORT uint64_t rc_compress_bound(uint32_t scheme, uint64_t n)
{
    const uint64_t blocks = (n + rc::TILE_BM - 1) / rc::TILE_BM;
    if (scheme == RC_SCHEME_LZ4) return 7 + n + 4 * blocks + 4;
    if (scheme == RC_SCHEME_ZSTD) return 9 + n + 3 * blocks;
    if (scheme == RC_SCHEME_BLOSC_LZ4) return 16 + n + 8 * blocks;
    return 0;
}
