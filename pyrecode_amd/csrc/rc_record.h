// rc_record.h - the byte layout of a record's two streams (shared by the layout kernel and k_gather): stream framing of the device
// codecs, stored-chunk positions, the fixed fields of a record.  Reference: pyrecode/recode_writer.py:485-494,518-525,546-550 (record),
// lz4_Frame_format.md / RFC 8878 / the blosc1 chunk header for the containers.
#pragma once
#include "rc_launch.h"

namespace rc {

// Framing of the two per-frame streams for the device codecs.  The bitmap stream is [hdr][encoded blocks][end]; the pixel
// stream is stored ("raw") chunks: [hdr]{[chunk header][<= 2^chunk_shift bytes]}[end].
//   LZ4 frame  (lz4_Frame_format.md): 7-byte header, 4-byte block words (bit 31 = stored), 4-byte EndMark, 4 MiB chunks
//   zstd frame (RFC 8878):            6-byte header (magic, descriptor, window), 3-byte block headers, Last_Block bit on the
//                                     final block instead of an end mark, 128 KiB chunks (Block_Maximum_Size), >= 1 block
//   blosc1 chunk (scheme 8):          bitmap: 16-byte header + int32 bstarts[ntiles] + blocks; pixels: 16-byte header with the
//                                     "memcpyed" flag + the bytes (what c-blosc itself emits for incompressible input)
//   zlib stream (RFC 1950 / 1951):    2-byte header, deflate blocks (the map: a byte-aligned block pair per tile, rc_deflate_block.h; the
//                                     residuals: stored blocks of 32 KiB with 5-byte headers, BFINAL on the last, >= 1 block), Adler-32
//                                     of the uncompressed bytes, big-endian (summed up inside k_gather, written by k_zlib_finish behind it)
struct FrameFmt { uint32_t hdr, end, chunk_shift, chunk_hdr, min_chunks; };
__host__ __device__ inline FrameFmt frame_fmt(uint32_t emit)
{
    return emit == 1 ? FrameFmt{6, 0, 17, 3, 1} : emit == 8 ? FrameFmt{16, 0, 31, 0, 0} : emit == EMIT_DEFLATE ? FrameFmt{2, 4, 15, 5, 1} : FrameFmt{7, 4, 22, 4, 0};
}
// bytes in front of the encoded bitmap blocks
__host__ __device__ inline uint32_t bitmap_hdr(const FrameFmt &ff, uint32_t emit, uint32_t ntiles)
{
    return emit == 8 ? 16u + 4u * ntiles : ff.hdr;
}

__host__ __device__ inline uint32_t packed_bytes(uint32_t nnz, uint32_t depth)
{
    return depth == 16 ? nnz * 2u : (uint32_t)(((uint64_t)nnz * depth + 7) >> 3);
}
__host__ __device__ inline uint32_t stored_chunks(const FrameFmt &ff, uint32_t n)
{
    const uint32_t c = (n + (1u << ff.chunk_shift) - 1) >> ff.chunk_shift;
    return c < ff.min_chunks ? ff.min_chunks : c;
}
__host__ __device__ inline uint32_t stored_size(const FrameFmt &ff, uint32_t n)
{
    return ff.hdr + n + ff.chunk_hdr * stored_chunks(ff, n) + ff.end;
}

__device__ __forceinline__ void store_u32_le(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}


// position of packed-pixel byte b inside the pixel stream's frame (stored chunks)
__device__ __forceinline__ uint64_t stored_pos(const FrameFmt &ff, uint64_t b)
{
    return ff.hdr + ff.chunk_hdr * ((b >> ff.chunk_shift) + 1) + b;
}


// The fixed fields of frame f's record (ids, sizes, the two streams' frame headers / end marks / stored-chunk headers): written by ONE lane.
template <class S>
__device__ __forceinline__ void record_fixed_fields(const S &sc, const RecordParams &rp, uint32_t f, uint8_t *rec, uint64_t bitmap_pos, uint64_t pix_pos,
                                                    uint32_t cb, uint32_t npk, const FrameFmt &ff, bool skip_pix, uint32_t lz4f_hdr_bitmap, uint32_t lz4f_hdr_pix)
{
    store_u32_le(rec, rp.first_frame_id + f);
    if (rp.emit == 0) {
        if (rp.level == 1) store_u32_le(rec + 4, npk);
    } else {
        store_u32_le(rec + 4, cb);
        uint8_t *bf = rec + bitmap_pos;
        if (rp.emit == 1) {  // zstd: magic, Frame_Header_Descriptor 0 (no content size, window descriptor follows), 1 KiB window
            store_u32_le(bf, 0xFD2FB528u);
            bf[4] = 0; bf[5] = 0x00;
        } else if (rp.emit == 8) {  // blosc1 header: version 2, LZ4 format version 1, bit-shuffle | not split | LZ4, typesize 8
            bf[0] = 2; bf[1] = 1; bf[2] = 0x34; bf[3] = 8;
            store_u32_le(bf + 4, (uint32_t)sc.nb);
            store_u32_le(bf + 8, (uint32_t)min((uint64_t)TILE_BM, sc.nb));
            store_u32_le(bf + 12, cb);
        } else if (rp.emit == EMIT_DEFLATE) {  // zlib: CMF = deflate with a 32 KiB window, FLG = fastest level + check bits
            bf[0] = 0x78; bf[1] = 0x01;
        } else {
            store_u32_le(bf, 0x184D2204u);
            bf[4] = (uint8_t)(lz4f_hdr_bitmap & 0xFF); bf[5] = (uint8_t)((lz4f_hdr_bitmap >> 8) & 0xFF);
            bf[6] = (uint8_t)((lz4f_hdr_bitmap >> 16) & 0xFF);
            store_u32_le(bf + cb - ff.end, 0);
        }
        if (rp.level == 1 && skip_pix) {
            store_u32_le(rec + 8, ff.hdr + sc.frame_pbytes[f]);
            store_u32_le(rec + 12, npk);
        } else if (rp.level == 1) {
            const uint32_t cp = stored_size(ff, npk);
            store_u32_le(rec + 8, cp);
            store_u32_le(rec + 12, npk);
            uint8_t *pf = rec + pix_pos;
            const uint32_t chunk = 1u << ff.chunk_shift, nch = stored_chunks(ff, npk);
            if (rp.emit == 8) {  // blosc1 header with the "memcpyed" flag: the packed residuals follow unchanged
                pf[0] = 2; pf[1] = 1; pf[2] = 0x36; pf[3] = 8;
                store_u32_le(pf + 4, npk);
                store_u32_le(pf + 8, npk);
                store_u32_le(pf + 12, 16 + npk);
            } else if (rp.emit == EMIT_DEFLATE) {  // stored blocks: [BFINAL][LEN][NLEN]
                pf[0] = 0x78; pf[1] = 0x01;
                for (uint32_t k = 0; k < nch; ++k) {
                    const uint32_t o = k * chunk, len = npk > o ? min(chunk, npk - o) : 0u;
                    uint8_t *q = pf + ff.hdr + (uint64_t)k * (chunk + 5);
                    q[0] = k + 1 == nch ? 1 : 0;
                    q[1] = (uint8_t)len; q[2] = (uint8_t)(len >> 8); q[3] = (uint8_t)~len; q[4] = (uint8_t)(~len >> 8);
                }
            } else if (rp.emit == 1) {  // 128 KiB window so that 128 KiB raw blocks are legal
                store_u32_le(pf, 0xFD2FB528u);
                pf[4] = 0; pf[5] = 7u << 3;
                for (uint32_t k = 0; k < nch; ++k) {
                    const uint32_t o = k * chunk, len = npk > o ? min(chunk, npk - o) : 0u;
                    const uint32_t h = (k + 1 == nch ? 1u : 0u) | (len << 3);  // Raw_Block
                    uint8_t *q = pf + ff.hdr + (uint64_t)k * (chunk + 3);
                    q[0] = (uint8_t)h; q[1] = (uint8_t)(h >> 8); q[2] = (uint8_t)(h >> 16);
                }
            } else {
                store_u32_le(pf, 0x184D2204u);
                pf[4] = (uint8_t)(lz4f_hdr_pix & 0xFF); pf[5] = (uint8_t)((lz4f_hdr_pix >> 8) & 0xFF);
                pf[6] = (uint8_t)((lz4f_hdr_pix >> 16) & 0xFF);
                for (uint32_t k = 0; k < nch; ++k) {
                    const uint32_t o = k * chunk, len = min(chunk, npk - o);
                    store_u32_le(pf + ff.hdr + (uint64_t)k * (chunk + 4), len | 0x80000000u);
                }
                store_u32_le(pf + cp - ff.end, 0);
            }
        }
    }
}

}  // namespace rc
