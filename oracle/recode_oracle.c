/* ORACLE - test infrastructure, NOT the product.
 *
 * Plain-C CPU restatement of pyReCoDe's per-frame reduce -> bit-pack hot path and of the reader's
 * sparse-expand, one function per reference stage, each citing the reference file:line it follows
 * (paths relative to /root/reference).  Pinned against golden vectors captured by running the
 * reference itself in the build container (tests/golden/make_golden.py -> tests/golden/ npz files,
 * checked by tests/test_oracle_golden.py) and, where oracle/_ref is present, against the
 * reference's own compiled C loops (oracle/_ref/libreader_ref.so).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * pyrecode_amd/ never does: the product path is the HIP library and fails loudly without it.
 *
 * Build: make -C oracle   (gcc -O3 -shared -fPIC -> oracle/librecode_oracle.so)
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* A1  thr = dark + epsilon, kept in the source dtype: NumPy-2 uint16 + python-int addition wraps
 *     mod 2^16.  pyrecode/recode_writer.py:126-127 (cast at :133-137 is a no-op for uint16 dark). */
ORC_API void orc_threshold(const uint16_t *dark, int64_t eps, uint64_t n, uint16_t *thr)
{
    for (uint64_t k = 0; k < n; ++k)
        thr[k] = (uint16_t)((uint64_t)dark[k] + (uint64_t)eps);
}

/* A2  binary = frame > thr (strict, uint16 compare).  pyrecode/recode_writer.py:437
 * A3  pix = frame[binary] - thr[binary], row-major order of the set pixels.  recode_writer.py:440
 * binary is one byte per pixel like numpy bool; returns nnz. */
ORC_API uint64_t orc_binarize_l1(const uint16_t *frame, const uint16_t *thr, uint64_t n,
                                 uint8_t *binary, uint16_t *pix)
{
    uint64_t nnz = 0;
    for (uint64_t k = 0; k < n; ++k) {
        uint8_t b = frame[k] > thr[k];
        binary[k] = b;
        if (b)
            pix[nnz++] = (uint16_t)(frame[k] - thr[k]);
    }
    return nnz;
}

/* A4  bit k%8 of byte k/8 <- binary.flat[k]; ceil(n/8) bytes, tail bits zero.
 *     pyrecode/recode_writer.py:622-634 (numba _pack_binary_frame), output length from :219. */
ORC_API void orc_pack_binary_frame(const uint8_t *binary, uint64_t n, uint8_t *out)
{
    uint64_t nb = (n + 7) / 8;
    memset(out, 0, nb);
    uint64_t count = 0;
    unsigned index = 0;
    for (uint64_t k = 0; k < n; ++k) {
        if (binary[k] == 1)
            out[count] |= (uint8_t)(1u << index);
        if (++index == 8) {
            ++count;
            index = 0;
        }
    }
}

/* A5  low d bits of each value, LSB first, value after value, no padding; ceil(n*d/8) bytes,
 *     zeroed first; bits >= d dropped.  pyrecode/recode_writer.py:637-652 (numba _bit_pack) =
 *     intended semantics of c_extensions/reader.h:105-140.  Returns bytes written. */
ORC_API uint64_t orc_bit_pack(const uint16_t *vals, uint64_t n, unsigned d, uint8_t *out)
{
    uint64_t n_packed = (n * d + 7) / 8;
    memset(out, 0, n_packed);
    uint64_t j = 0;
    unsigned bp = 0;
    for (uint64_t p = 0; p < n; ++p) {
        for (unsigned i = 0; i < d; ++i) {
            if (i < 16 && (vals[p] & (1u << i)))
                out[j] |= (uint8_t)(1u << bp);
            if (++bp == 8) {
                ++j;
                bp = 0;
            }
        }
    }
    return n_packed;
}

/* inverse of A5: n values of d bits each, LSB first -> uint64.  Intended semantics of
 * c_extensions/reader.h:74-99 (whose loop never terminates as shipped). */
ORC_API void orc_bit_unpack(const uint8_t *packed, uint64_t n, unsigned d, uint64_t *out)
{
    for (uint64_t v = 0; v < n; ++v) {
        uint64_t x = 0;
        for (unsigned b = 0; b < d; ++b) {
            uint64_t k = v * d + b;
            if (packed[k / 8] & (1u << (k % 8)))
                x |= 1ull << b;
        }
        out[v] = x;
    }
}

/* A10 sparse expand: for every set bit of the bitmap in row-major order emit (row, col, val);
 *     level 1: val = next d-bit LSB-first field of pix; other levels: val = 1.  Returns nnz.
 *     pyrecode/c_extensions/reader.h:10-68, called from pyrecode.cpp:95-119. */
ORC_API int64_t orc_unpack_frame_sparse(uint32_t nx, uint32_t ny, unsigned d, const uint8_t *bitmap,
                                        const uint8_t *pix, uint64_t *out, unsigned level)
{
    uint64_t nfg = 0;
    for (uint32_t row = 0; row < ny; ++row) {
        for (uint32_t col = 0; col < nx; ++col) {
            uint64_t k = (uint64_t)row * nx + col;
            if (!(bitmap[k / 8] & (1u << (k % 8))))
                continue;
            uint64_t val = 1;
            if (level == 1) {
                val = 0;
                for (unsigned b = 0; b < d; ++b) {
                    uint64_t q = nfg * d + b;
                    if (pix[q / 8] & (1u << (q % 8)))
                        val |= 1ull << b;
                }
            }
            out[nfg * 3] = row;
            out[nfg * 3 + 1] = col;
            out[nfg * 3 + 2] = val;
            ++nfg;
        }
    }
    return (int64_t)nfg;
}

/* ------------------------------------------------------------------------------------------
 * Fused single-pass form of A2+A3+A4+A5 for the timed CPU baseline ("port" kind in bench.py):
 * same results as the stage functions above (tests/test_oracle_golden.py checks that), written
 * the way a competent -O3 CPU port would be - one read of frame and thr, 8 pixels per bitmap
 * byte, 64-bit bit accumulator for the d-bit fields.  recode_writer.py:437-475.
 * bitmap: ceil(n/8) bytes; packed: ceil(nnz*d/8) bytes (raw LE uint16 when d == 16).
 * Returns nnz; *n_packed = bytes written to packed.
 * ------------------------------------------------------------------------------------------ */
ORC_API uint64_t orc_reduce_frame_l1(const uint16_t *frame, const uint16_t *thr, uint64_t n, unsigned d,
                                     uint8_t *bitmap, uint8_t *packed, uint64_t *n_packed)
{
    uint64_t nnz = 0, acc = 0, j = 0;
    unsigned nacc = 0;
    const uint64_t vmask = d >= 16 ? 0xFFFFull : ((1ull << d) - 1);
    uint64_t nfull = n / 8;
    for (uint64_t g = 0; g < nfull; ++g) {
        const uint16_t *f = frame + g * 8, *t = thr + g * 8;
        unsigned m = 0;
        for (unsigned i = 0; i < 8; ++i)
            m |= (unsigned)(f[i] > t[i]) << i;
        bitmap[g] = (uint8_t)m;
        while (m) {
            unsigned i = (unsigned)__builtin_ctz(m);
            m &= m - 1;
            acc |= ((uint64_t)(uint16_t)(f[i] - t[i]) & vmask) << nacc;
            nacc += d;
            ++nnz;
            while (nacc >= 8) {
                packed[j++] = (uint8_t)acc;
                acc >>= 8;
                nacc -= 8;
            }
        }
    }
    if (n % 8) {
        unsigned m = 0;
        for (uint64_t k = nfull * 8; k < n; ++k) {
            if (frame[k] > thr[k]) {
                m |= 1u << (k % 8);
                acc |= ((uint64_t)(uint16_t)(frame[k] - thr[k]) & vmask) << nacc;
                nacc += d;
                ++nnz;
                while (nacc >= 8) {
                    packed[j++] = (uint8_t)acc;
                    acc >>= 8;
                    nacc -= 8;
                }
            }
        }
        bitmap[nfull] = (uint8_t)m;
    }
    if (nacc)
        packed[j++] = (uint8_t)acc;
    *n_packed = j;
    return nnz;
}

/* ------------------------------------------------------------------------------------------
 * Stock-format DECODERS written from the published format documents, used by tests to prove that
 * the device-emitted streams are valid and expand to the bit-exact payload even on a box without
 * liblz4 (tests also cross-check with the system liblz4.so.1 when it is there).
 *
 * LZ4 frame format v1.6.x (lz4_Frame_format.md) + LZ4 block format (lz4_Block_format.md).
 * Handles: magic 0x184D2204, FLG/BD, optional content size / dict id, block-independent or linked
 * blocks (history = everything decoded so far in dst), uncompressed blocks (high bit of block
 * size), block checksums skipped, EndMark, content checksum skipped.
 * Returns decoded byte count, or negative on malformed input / overflow of dst_cap.
 * Reference call site whose inverse this is: pyrecode/recode_compressors.py:49 (lz4.frame.decompress).
 * ------------------------------------------------------------------------------------------ */
static int64_t lz4_block_decode(const uint8_t *src, uint64_t n, uint8_t *dst_base, uint64_t dst_pos,
                                uint64_t dst_cap)
{
    uint64_t ip = 0, op = dst_pos;
    while (ip < n) {
        unsigned token = src[ip++];
        uint64_t lit = token >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (ip >= n) return -10;
                b = src[ip++];
                lit += b;
            } while (b == 255);
        }
        if (ip + lit > n || op + lit > dst_cap) return -11;
        memcpy(dst_base + op, src + ip, lit);
        ip += lit;
        op += lit;
        if (ip >= n) break; /* last sequence: literals only */
        if (ip + 2 > n) return -12;
        uint64_t off = src[ip] | ((uint64_t)src[ip + 1] << 8);
        ip += 2;
        if (off == 0 || off > op) return -13;
        uint64_t ml = token & 15;
        if (ml == 15) {
            unsigned b;
            do {
                if (ip >= n) return -14;
                b = src[ip++];
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (op + ml > dst_cap) return -15;
        for (uint64_t i = 0; i < ml; ++i, ++op) /* byte-wise: overlapping copies are the RLE case */
            dst_base[op] = dst_base[op - off];
    }
    return (int64_t)(op - dst_pos);
}

/* raw LZ4 block (no frame), as stored inside blosc1 chunks; returns decoded bytes or negative */
ORC_API int64_t orc_lz4_block_decode(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap)
{
    return lz4_block_decode(src, n, dst, 0, dst_cap);
}

ORC_API int64_t orc_lz4f_decode(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap)
{
    if (n < 7) return -1;
    uint32_t magic = src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24);
    if (magic != 0x184D2204u) return -2;
    unsigned flg = src[4], bd = src[5];
    if ((flg >> 6) != 1) return -3;            /* version must be 01 */
    if (flg & 0x02) return -3;                 /* reserved bit */
    if (bd & 0x8F) return -3;                  /* reserved bits */
    unsigned bmax_code = (bd >> 4) & 7;
    if (bmax_code < 4) return -3;
    uint64_t bmax = 1ull << (8 + 2 * bmax_code); /* 4->64KB 5->256KB 6->1MB 7->4MB */
    int b_checksum = (flg >> 4) & 1, c_size = (flg >> 3) & 1, c_checksum = (flg >> 2) & 1, dict_id = flg & 1;
    uint64_t ip = 6 + (c_size ? 8 : 0) + (dict_id ? 4 : 0) + 1; /* +1 header checksum byte (not verified) */
    uint64_t op = 0;
    for (;;) {
        if (ip + 4 > n) return -4;
        uint32_t bs = src[ip] | (src[ip + 1] << 8) | (src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
        ip += 4;
        if (bs == 0) break; /* EndMark */
        int raw = bs >> 31;
        bs &= 0x7FFFFFFFu;
        if (bs > bmax || ip + bs > n) return -5;
        if (raw) {
            if (op + bs > dst_cap) return -6;
            memcpy(dst + op, src + ip, bs);
            op += bs;
        } else {
            int64_t got = lz4_block_decode(src + ip, bs, dst, op, dst_cap);
            if (got < 0) return got;
            if ((uint64_t)got > bmax) return -7;
            op += (uint64_t)got;
        }
        ip += bs + (b_checksum ? 4 : 0);
    }
    if (c_checksum) ip += 4;
    if (ip != n) return -8; /* trailing garbage: n_comp_* metadata must equal the stream length */
    return (int64_t)op;
}
