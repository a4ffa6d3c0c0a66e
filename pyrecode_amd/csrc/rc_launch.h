// rc_launch.h - host-side launcher declarations shared by the .hip translation units of librecode_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rc_device.h"

namespace rc {

// Device scratch of one ctx, laid out for a batch of up to max_batch frames (DESIGN.md "data layout in HBM").
struct Scratch {
    uint64_t N = 0;            // pixels per frame
    uint32_t ntiles = 0;       // ceil(N / TILE_PX)
    uint64_t nb = 0;           // ceil(N / 8) bitmap bytes per frame
    uint64_t nb_stride = 0;    // ntiles * TILE_BM (bitmap rows padded to whole tiles)
    uint32_t max_batch = 0;
    bool guarded_loads = false;        // every tile through the guarded single-load instantiation of the reduce kernel (tests: RC_REDUCE_GUARDED_LOADS=1
                                       // in the environment when the ctx is created; the product reads the switch nowhere else)
    uint16_t *thr = nullptr;           // [N]
    uint8_t *bitmap = nullptr;         // [B][nb_stride]            packed binary maps
    uint16_t *pix_slots = nullptr;     // [B][ntiles][TILE_PX]      per-tile residuals, row-major inside the tile
    uint32_t pix_slot_bytes = SLOT_PX * 2;   // bytes between two tiles' residual slots (uint32 sources, rc_reduce32.hip: TILE_PX * 4)
    uint32_t *tile_cnt = nullptr;      // [B][ntiles]               set pixels per tile
    uint32_t *tile_off = nullptr;      // [B][ntiles]               exclusive prefix of tile_cnt inside the frame
    uint32_t *tile_next = nullptr;     // [B][ntiles]               next tile index > t with tile_cnt > 0 (ntiles if none)
    uint8_t *blk_slots = nullptr;      // [B][ntiles][blk_stride]   encoded bitmap blocks (codec dependent); a ctx's level-1 pipeline keeps the
                                       //                           tile's residual stream in the same slot when both fit (`comb`)
    uint32_t blk_stride = BLK_SLOT;    // bytes between two tiles' block slots
    // Combined slots (a ctx with a device codec at level 1): the tile's packed residual stream sits BEHIND its encoded block in the block's
    // slot - 1: at the next 16-byte boundary behind the block image (LZ4 / blosc: the image is final when the reduce kernel writes it),
    // 2: at offset BLK_SLOT (zstd: the FSE pass and the frame's definitions still grow the block) - whenever block + residual lines fit
    // blk_stride; a tile too dense for that keeps its residuals in pix_slots as before.  One run of lines per tile and frame instead of
    // two runs 8 KiB apart: fewer lines written by the reduce kernel and read by k_gather (each run wastes half a line on average).
    uint32_t comb = 0;
    uint32_t *blk_size = nullptr;      // [B][ntiles]               bytes used in each slot
    uint32_t *blk_aux = nullptr;       // [B][ntiles]               deflate: the tiles' Adler-32 partials (rc_deflate_block.h::deflate_adler_word)
    uint32_t *zl_acc = nullptr;        // [B][8]                    deflate: per frame {A, W of the map, A, W of the residual stream}
                                       //                           (zeroed by k_layout, summed up by k_gather, turned into the two trailers by k_zlib_finish)
    uint32_t *blk_off = nullptr;       // [B][ntiles]               exclusive prefix of blk_size inside the frame
    uint32_t *frame_nnz = nullptr;     // [B]
    uint32_t *frame_cbytes = nullptr;  // [B]                       sum of blk_size
    uint32_t *scan_part = nullptr;     // [B][nseg][8]              partial results of the segmented scans (frames with > 4096 tiles)
    BatchStatus *status = nullptr;     // [1] of this batch
    BatchStatus *first_err = nullptr;  // [1] shared by both scratch sets: first failed batch since the last rc_ctx_sync
                                       //     (code, frame, total = number of the batch among those enqueued since then)
    // modelled zstd, residual stream (rc_pix_huff.hip): the frame's packed stream flat, its PIX_CHUNK-byte chunks encoded
    uint8_t *pixraw = nullptr;         // [B][pixraw_stride]
    uint64_t pixraw_stride = 0;
    uint8_t *pix_chunks = nullptr;     // [B][nchunk_max][PIX_SLOT]
    uint32_t *chunk_size = nullptr;    // [B][nchunk_max]   encoded bytes (ZW_TREE flag: the chunk's literals are treeless)
    uint32_t *chunk_off = nullptr;     // [B][nchunk_max]   exclusive prefix inside the frame
    uint32_t *frame_pbytes = nullptr;  // [B]               sum of chunk_size
    uint32_t nchunk_max = 0;
    // modelled zstd (codec 3, rc_zstd_model.h): the ctx's model on the device and what the block encoders need of it
    const void *zm_model = nullptr;    // ZstdModel
    const void *zm_lit_code = nullptr; // &model->lit_code
    uint32_t zm_valid = 0, zm_budget = 0, zm_seq_bits = 12;
    // level 2 (rc_l2.hip)
    u32x2 *l2_node = nullptr;          // [B][ntiles * TILE_PX] {parent id, accumulator} per set pixel, id = tile * TILE_PX + rank in the tile
    uint64_t l2_ids_per_frame = 0;
    uint16_t *l2_base = nullptr;       // [B][ntiles * 64] set pixels of a word's tile in front of the word (k_l2_dir)
};

// RecordParams::emit / rc_ctx::emit of the device DEFLATE encoder (include/recode_hip.h: RC_SCHEME_ZLIB_DEVICE); the others are the
// reference's compression_scheme codes
constexpr uint32_t EMIT_DEFLATE = 0x100u;
struct RecordParams {
    uint32_t level;        // 1 or 3
    uint32_t emit;         // 0 = raw pieces (mode-0 record), 2 = LZ4 frames, 1 = zstd frames, 8 = blosc-lz4, EMIT_DEFLATE = zlib streams
    uint32_t depth;        // source_bit_depth
    uint32_t packed_slots; // 1: the tiles' slots hold tile-local packed streams - level-1 residuals and, since round 5, level-2 statistics (k_l2_emit);
                           // 0: no value stream is gathered (no caller passes it any more)
    uint32_t first_frame_id;
    uint64_t frame_bytes;  // raw frame size = N * 2 (record upper bound, recode_writer.py:565-566)
    uint32_t pix_mode = 0; // k_gather / k_layout: 0 = the residual stream goes into the record as it is (stored chunks);
                           // 1 = ONLY the residual stream, flat, into Scratch::pixraw (input of the Huffman stage);
                           // 2 = everything but the residual stream, whose encoded size is Scratch::frame_pbytes
};

// rc_reduce.hip
void launch_threshold(const void *dark, int64_t eps, uint64_t N, uint16_t *thr, hipStream_t s, uint32_t src_bytes = 2);   // dark: uint16, or uint8 for src_bytes 1
// codec: 0 none, 2 LZ4, 1 zstd (plain), 3 zstd (modelled), 8 blosc-lz4, 5 deflate (fixed-Huffman block per tile).
// level: 1 residuals, 2 raw values of the set pixels (input of launch_l2), 3 bitmap only.  depth < 16 (level 1 only):
// every tile's residuals are left in its slot already bit-packed (tile-local LSB-first stream of depth-bit fields)
// src_bytes: bytes per source pixel - 2 (uint16 frames) or 1 (uint8 frames, source_bit_depth <= 8)
void launch_reduce(const Scratch &sc, const void *frames, uint32_t B, uint32_t level, uint32_t codec, bool keep_bitmap,
                   uint32_t depth, hipStream_t s, hipStream_t s_tail = nullptr, uint32_t src_bytes = 2);
// rc_reduce32.hip: uint32 sources (source_bit_depth > 16) - reduce + d-bit pack, the block encoder of `codec` fused (2 / 4 LZ4 runs / events,
// 8 blosc-lz4, 1 zstd fast form); raw binary maps only with keep_bitmap
void launch_threshold32(const uint32_t *dark, int64_t eps, uint64_t N, uint32_t *thr, hipStream_t s);
void launch_reduce32(const Scratch &sc, const uint32_t *frames, const uint32_t *thr32, uint32_t B, uint32_t level, uint32_t depth, hipStream_t s,
                     uint32_t codec = 0, bool keep_bitmap = true);   // codec 2 / 4: the LZ4 block encoder (runs / events) fused
// rc_l2.hip
void launch_l2(const Scratch &sc, uint32_t B, uint32_t nx, uint32_t use_sum, uint32_t depth, hipStream_t s);
void launch_scans(const Scratch &sc, uint32_t B, bool with_counts, bool with_blocks, hipStream_t s);
void launch_layout(const Scratch &sc, const RecordParams &rp, uint32_t B, uint64_t out_cap, uint64_t *rec_off,
                   uint32_t *md, hipStream_t s);
void launch_assemble(const Scratch &sc, const RecordParams &rp, uint32_t B, uint8_t *out, const uint64_t *rec_off,
                     uint32_t batch_seq, hipStream_t s);
// rc_gather.hip: k_gather, what launch_assemble runs for everything but level-2 value lists
void launch_gather(const Scratch &sc, const RecordParams &rp, uint32_t B, uint8_t *out, const uint64_t *rec_off, uint32_t hdr_bitmap, uint32_t hdr_pix,
                   uint32_t batch_seq, hipStream_t s);
// rc_pix_huff.hip: the packed residual stream of every frame (Scratch::pixraw) -> Huffman-coded zstd blocks -> the records
constexpr uint32_t PIX_CHUNK = 1008, PIX_SLOT = 1024;
void launch_pix_huff(const Scratch &sc, uint32_t B, uint32_t depth, hipStream_t s);
void launch_pix_scan(const Scratch &sc, uint32_t B, uint32_t depth, hipStream_t s);   // (rc_reduce.hip)
void launch_pix_gather(const Scratch &sc, uint32_t B, uint32_t depth, uint32_t level1_hdr, uint8_t *out, const uint64_t *rec_off,
                       hipStream_t s);
// rc_lz4.hip
struct Lz4Block { uint64_t src_off; uint32_t size; uint32_t raw; };
void launch_lz4_encode_buffer(const Scratch &sc, hipStream_t s, bool events = false);  // sc.bitmap = the buffer, sc.nb = its length
void launch_lz4_encode_rows(const Scratch &sc, uint32_t B, hipStream_t s, bool events);   // B rows of sc.bitmap (a batch's raw binary maps)
void launch_lz4f_gather(const Scratch &sc, uint32_t hdr3, uint8_t *out, hipStream_t s);
void launch_lz4_decode(const uint8_t *src, const Lz4Block *blks, uint32_t nblk, uint32_t *sizes, const uint64_t *dst_off,
                       uint8_t *dst, uint64_t cap, int linked, int *err, hipStream_t s, uint32_t max_stored = 0);
uint32_t lz4f_descriptor(uint8_t bd);
// rc_blosc.hip
void launch_blosc_encode_blocks(const Scratch &sc, uint32_t B, hipStream_t s);
void launch_blosc_gather(const Scratch &sc, uint8_t *out, hipStream_t s);
void launch_blosc_unshuffle(const uint8_t *in, uint8_t *out, uint64_t nbytes, uint32_t blocksize, uint32_t typesize,
                            uint32_t shuffle, hipStream_t s);
// rc_zstd.hip
void launch_zstd_encode_blocks(const Scratch &sc, uint32_t B, const void *tables_dev, hipStream_t s);
void launch_zstd_tokenize_rows(const Scratch &sc, uint32_t B, hipStream_t s);   // the tokenizer half only (launch_zstd_fse finishes the blocks)
void launch_zstd_fse(const Scratch &sc, uint32_t B, const void *tables_dev, bool fitted, hipStream_t s);  // 2nd half of the fused path
// modelled encoder (rc_zstd_model.h): histograms of a sample of plain-tokenized frames -> model (host) -> kernels
void launch_zstd_sample(const Scratch &sc, uint32_t B, bool with_pix, uint32_t depth, void *sample_dev, hipStream_t s);
size_t zstd_model_bytes();
size_t zstd_sample_bytes();
void zstd_model_from_sample(const void *sample_host, void *model_host, uint32_t speed_permille = 0);   // (rc_zstd_model.h::zm_build_model)
struct ZstdModel;
void launch_zstd_gather(const Scratch &sc, uint8_t *out, hipStream_t s);
size_t zstd_tables_bytes();
void zstd_tables_host(void *dst);  // rc_reduce.hip: FLG | BD << 8 | HC << 16

// where tile ft's packed residual stream starts (see Scratch::comb); bn: the tile's blk_size word as the reduce kernel wrote it
// (combined form 1 only), cnt: its set pixels, d: bits per value
template <class S>
__host__ __device__ inline const uint8_t *residual_src(const S &sc, uint64_t ft, uint32_t bn, uint32_t cnt, uint32_t d)
{
    if (sc.comb) {
        const uint32_t ro16 = sc.comb == 2 ? (uint32_t)BLK_SLOT / 16 : (bn + 15) >> 4, r16 = (cnt * d + 127) >> 7;
        if (16 * (ro16 + r16) <= sc.blk_stride) return sc.blk_slots + ft * sc.blk_stride + 16 * ro16;
    }
    return reinterpret_cast<const uint8_t *>(sc.pix_slots) + ft * sc.pix_slot_bytes;
}

void launch_roi_components(const void *frames, const void *thr, uint64_t N, uint32_t nx, uint32_t n, uint32_t first_frame_id, uint32_t gap,
                           uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t *counts, hipStream_t s, uint32_t src_bytes = 2);
}  // namespace rc
