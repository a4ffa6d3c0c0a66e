// rc_expand.h - launchers of rc_expand.hip
#pragma once
#include "rc_device.h"

namespace rc {
void launch_expand_count(const uint8_t *bitmap_pad8, uint64_t nb8, uint64_t N, uint32_t *blk_cnt, uint32_t *blk_off,
                         uint64_t *nnz_dev, hipStream_t s);
void launch_expand_emit(const uint8_t *bitmap_pad8, uint64_t nb8, uint64_t N, uint32_t nx, const uint32_t *blk_off,
                        const uint8_t *pix, uint64_t pix_bytes, uint32_t d, uint32_t level, uint64_t cap, uint64_t *out,
                        hipStream_t s);
void launch_expand_batch_count(const uint8_t *bm, uint64_t bm_stride, uint64_t nb8, uint64_t N, uint32_t n, uint32_t *blk_cnt, uint32_t *blk_off,
                               uint64_t *frame_nnz, uint64_t *frame_base, hipStream_t s, const uint32_t *pv_bytes = nullptr, uint32_t d = 0,
                               uint32_t level = 0, uint64_t cap = 0, int *err = nullptr);
void launch_expand_batch_emit(const uint8_t *bm, uint64_t bm_stride, uint64_t nb8, uint64_t N, uint32_t nx, uint32_t n, const uint32_t *blk_off,
                              const uint64_t *frame_base, const uint8_t *pv, uint64_t pv_stride, const uint32_t *pv_bytes, uint32_t d,
                              uint32_t level, uint64_t cap, void *out, hipStream_t s, const int *err = nullptr, bool coo = false);
// rc_zstd_dec.hip: block decoders of the batched reader
void launch_block_decode(int codec, int row, const uint8_t *data, const void *frame_lists, uint32_t nframes, uint32_t max_blocks_per_frame,
                         const void *tables, const void *predef, uint8_t *out, const uint64_t *out_base, int *err, hipStream_t s,
                         uint32_t *produced_out = nullptr);
void launch_bitmap_decode_compact(int codec, const uint8_t *data, const void *frame_lists, const uint64_t *src_base, uint32_t nframes,
                                  uint32_t max_blocks_per_frame, const void *tables, const void *predef, uint8_t *out, const uint64_t *out_base,
                                  uint64_t nb, int *err, hipStream_t s);
void launch_block_copy(const uint8_t *data, const void *lists, uint32_t nlists, uint32_t nblocks, uint32_t max_regen, uint8_t *out,
                       const uint64_t *out_base, hipStream_t s);
size_t zd_tables_bytes();
size_t zd_block_bytes();
void zd_predefined_tables(void *dst);
void launch_bit_pack(const uint16_t *vals, uint64_t n, uint32_t d, uint8_t *out, uint64_t out_n, hipStream_t s);
void launch_bit_unpack(const uint8_t *packed, uint64_t nbytes, uint64_t n, uint32_t d, uint64_t *out, hipStream_t s);
void launch_synth_dark(uint32_t seed, uint64_t N, uint16_t *dark, hipStream_t s);
void launch_synth_frames_clustered(uint32_t seed, uint32_t first_frame, uint32_t nframes, uint32_t nx, uint32_t ny, uint32_t seed_ppm,
                                   const uint16_t *dark, uint16_t *frames, hipStream_t s);
void launch_synth_frames(uint32_t seed, uint32_t first_frame, uint32_t nframes, uint64_t N, uint32_t sparsity_ppm,
                         const uint16_t *dark, uint16_t *frames, hipStream_t s);
}  // namespace rc
