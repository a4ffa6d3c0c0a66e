/* recode_hip.h - C ABI of librecode_hip.so: the MI355X (gfx950) implementation of pyReCoDe's per-frame
 * reduce -> bit-pack -> compress hot path and of the reader's sparse expand.
 *
 * This is the drop-in boundary.  Plain C: pointers and sizes only, no torch / HIP types.
 * Every entry point names the reference interface it replaces (paths relative to the reference repo).
 * All kernels behind it are hand-written HIP for gfx950; there is NO CPU fallback in this library: on a
 * machine without a usable GPU every compute entry point returns RC_ERR_DEVICE.
 *
 * Conventions
 *   - Return value: RC_OK (0) or a negative rc_status.  Nothing throws across the ABI.
 *   - Buffers are caller-owned (as in the reference: recode_writer.py:229-230, recode_reader.py:115).
 *     A data pointer may be host memory or device memory of the ctx's GPU; the library detects which
 *     (hipPointerGetAttributes) and stages host buffers itself.  It never allocates or frees caller memory.
 *   - A ctx is bound to one GPU and one HIP stream and is not thread-safe; distinct ctxs are independent
 *     (reference: one ReCoDeWriter per process, recode_server.py:358-363).
 *   - All multi-byte fields the library writes are little-endian (reference uses sys.byteorder on x86).
 */
#ifndef RECODE_HIP_H
#define RECODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 3

typedef enum rc_status {
    RC_OK = 0,
    RC_ERR_BAD_ARG = -1,          /* python shim: ValueError */
    RC_ERR_OUT_TOO_SMALL = -2,    /* caller's output capacity insufficient; nothing useful written */
    RC_ERR_DEVICE = -3,           /* no GPU / HIP error; rc_last_error() has the HIP message; python: RuntimeError */
    RC_ERR_UNSUPPORTED = -4,      /* scheme / level not implemented on device; python: NotImplementedError
                                     (recode_compressors.py:78-79,119-120) */
    RC_ERR_RECORD_TOO_LARGE = -5, /* a record exceeds the raw frame size; python: ValueError('Buffer size smaller
                                     than compressed data size') (recode_writer.py:565-566) */
    RC_ERR_CORRUPT = -6,          /* malformed compressed stream / bitmap-vs-pixvals mismatch on the read side */
    RC_ERR_WORKSPACE = -7         /* the device memory a ctx or a call needs could not be allocated (hipErrorOutOfMemory; rc_last_error
                                     names the allocation): python MemoryError.  A ctx's scratch is sized by its geometry and max_batch
                                     alone - per frame of the batch and scratch set (two sets): 2 * nx * ny bytes of residual slots +
                                     ceil(nx * ny / 4096) * 1.5 KiB of block slots (reduction level 2 adds 8 bytes per pixel and frame
                                     for the labelling nodes, a second copy of them once the ctx is pipelined) - so a ctx that was
                                     created never runs out of memory in a batch */
} rc_status;

/* compression_scheme codes of the reference (recode_compressors.py:3-4, config/README.md). Device codecs:
 * 2 (LZ4 frame), 1 (zstd frame), 8 (blosc1 chunk: bit-shuffle + LZ4, typesize 8).  Every other code: the ctx emits the
 * reduce-only pieces and the host layer runs the reference's own library call (zlib, bz2, lzma, ...).
 * RC_SCHEME_ZLIB_DEVICE is not a code of the reference: a ctx created with it writes compression_scheme 0 records (file header
 * field 0, recode_compressors.py:42-43,84-85) whose two streams are zlib streams (RFC 1950) made by the device's own DEFLATE
 * encoder - a byte-aligned fixed-Huffman block per 512-byte tile of the binary map, stored blocks for the packed values, Adler-32
 * trailers - which `zlib.decompress` (the reference's reader, recode_compressors.py:43) expands to the exact bytes; like every other
 * device codec it promises a VALID stream, not stock zlib's bytes (RC_SCHEME_ZLIB through the host layer keeps those). */
enum { RC_SCHEME_ZLIB = 0, RC_SCHEME_ZSTD = 1, RC_SCHEME_LZ4 = 2, RC_SCHEME_BLOSC_LZ4 = 8, RC_SCHEME_ZLIB_DEVICE = 0x100 };

typedef struct rc_ctx rc_ctx;

/* ---- library ------------------------------------------------------------------------------------- */
int rc_abi_version(void);
const char *rc_strerror(int status);
const char *rc_last_error(void);                 /* thread-local detail of the last failure */
int rc_device_count(int *count);                 /* RC_OK with *count == 0 when no GPU is visible */
int rc_scheme_on_device(uint32_t scheme);        /* 1 if rc_reduce_compress_batch emits this scheme itself */

/* ---- seam 1: the per-frame operator, batched ---------------------------------------------------------
 * Replaces ReCoDeWriter._reduce_compress (pyrecode/recode_writer.py:430-557) and the buffers ReCoDeWriter.start()
 * allocates for it (:212-230).  One ctx == one writer (one node_id).
 *
 *   nx, ny            frame shape (header fields nx, ny; recode_header.py:66-67)
 *   src_bit_depth     source_bit_depth, 1..32: pixvals are bit-packed when it is not a multiple of 8 (recode_writer.py:463-475);
 *                     up to 8 the reference's source dtype is uint8, beyond 16 uint32: rc_ctx_set_source_bytes(ctx, 1 / 4)
 *   reduction_level   1 (binary map + residuals), 2 (binary map + one statistic per 8-connected component, see
 *                     rc_ctx_set_l2_statistics) or 3 (binary map only); 4 -> RC_ERR_UNSUPPORTED
 *   op_mode           rc_operation_mode: 0 reduce only, 1 reduce + compress (recode_writer.py:482,497)
 *   scheme, clevel    compression_scheme / compression_level (recode_writer.py:503-511)
 *   device_id         HIP device ordinal
 *   max_batch         largest n later passed to rc_reduce_compress_batch (1 .. 65535: a batch's frames are a launch's grid.y); sizes the device scratch
 */
rc_ctx *rc_ctx_create(uint32_t nx, uint32_t ny, uint32_t src_bit_depth, uint32_t reduction_level,
                      uint32_t op_mode, uint32_t scheme, uint32_t clevel, int device_id, uint32_t max_batch,
                      int *status);
int rc_ctx_destroy(rc_ctx *ctx);

/* Reduction level 2 only: which statistic of the RAW frame values each connected component contributes (header field
 * L2_statistics, recode_writer.py:358-365): 0 or 1 = maximum, 2 = sum modulo 2^src_bit_depth - the reference casts the
 * statistic to the source dtype and stores it in src_bit_depth bits like every pixel value (recode_writer.py:446,463-475),
 * so a sum that does not fit wraps.
 * Components are listed in scipy.ndimage.label order, i.e. by their first pixel in row-major order
 * (recode_writer.py:443-446, utils/converters.py:262-297 - restated by intent, the reference's own code cannot run). */
int rc_ctx_set_l2_statistics(rc_ctx *ctx, uint32_t l2_statistics);

/* Use a caller-provided hipStream_t (passed as void*) instead of the ctx's own stream, e.g. torch's current
 * stream so that caller-side events bracket the kernels.  NULL restores the ctx's own stream (so the legacy null
 * stream, whose handle is 0, cannot be selected: use a created stream). */
int rc_ctx_set_stream(rc_ctx *ctx, void *hip_stream);

/* Bytes per SOURCE pixel - the reference's Python path takes whatever its map_dtype yields for the source bit depth (misc.py:41-49; only
 * use_c is uint16-only, recode_writer.py:85-87): 2 (default: uint16 frames and dark, 9..16 bits), 1 (uint8, <= 8 bits) or 4 (uint32, 17..32 bits).
 * Call before rc_set_dark and the first batch.  Every `frames` / `dark` pointer below is then of that type, a raw frame is ny*nx*bytes
 * (rc_out_capacity, the record bound of recode_writer.py:565-566).  uint8: a uint8 instantiation of the load path (half the bytes); everything
 * behind the loads is the uint16 path.  uint32 (levels 1 and 3): a kernel of its own for the reduce step (rc_reduce32.hip: uint32 compare,
 * residuals and depth-bit fields - four raw bytes a value when the depth is a multiple of 8, 24 included, as `.tobytes()` gives them,
 * recode_writer.py:463-464) with every device codec's block encoder fused into it as in the uint16 kernel (LZ4, blosc-lz4, zstd in its fast
 * form - the modelled encoder and reduction level 2 are uint16 / uint8 only); scans, record layout and assembly are shared.  rc_set_threshold then takes a uint32 frame. */
int rc_ctx_set_source_bytes(rc_ctx *ctx, uint32_t bytes_per_pixel);
uint32_t rc_ctx_source_bytes(const rc_ctx *ctx);

/* thr = calibration frame + epsilon in the source dtype's arithmetic (wraps mod 2^16 - mod 2^8 for uint8 sources - like NumPy 2):
 * ReCoDeWriter.__init__, recode_writer.py:126-137.  dark: uint16[ny*nx] (uint8[ny*nx] after rc_ctx_set_source_bytes(ctx, 1)), C order. */
int rc_set_dark(rc_ctx *ctx, const void *dark, int64_t epsilon);
/* Or hand over the finished threshold frame (self._calibration_frame_p_threshold). */
int rc_set_threshold(rc_ctx *ctx, const void *thr);   /* uint16[ny*nx] (uint8 and uint16 sources), uint32[ny*nx] (uint32 sources) */

/* Worst-case bytes one batch of n frames can occupy in `out` (n * raw frame size, the reference's own bound,
 * recode_writer.py:217-218,565-566), and the number of u32 metadata fields per frame for this ctx's
 * (level, mode) (structures.py:18-46): L1/mode1 3, L1/mode0 1, L3/mode1 1, L3/mode0 0. */
uint64_t rc_out_capacity(const rc_ctx *ctx, uint32_t n);
uint32_t rc_md_fields(const rc_ctx *ctx);

/* n frames in, n part-file records out.
 *   frames         uint16[n][ny][nx] C order (host or device); uint8[n][ny][nx] after rc_ctx_set_source_bytes(ctx, 1)
 *   first_frame_id absolute_frame_index of frames[0] (recode_writer.py:385); frame i gets first_frame_id + i
 *   out            records back to back, byte-identical in layout to what _write_to_frame_buffer assembles
 *                  (recode_writer.py:485-494,518-525,546-550):
 *                    L1 mode 1: u32 frame_id | u32 n_comp_bitmap | u32 n_comp_pix | u32 n_packed_pix | comp_bitmap | comp_pix
 *                    L1 mode 0: u32 frame_id | u32 n_packed_pix | bitmap[ceil(nx*ny/8)] | packed_pix
 *                    L3 mode 1: u32 frame_id | u32 n_comp_bitmap | comp_bitmap
 *                    L3 mode 0: u32 frame_id | bitmap
 *                  With op_mode 1 and a scheme that is not a device codec the ctx emits the mode-0 record and the
 *                  host layer compresses (see rc_scheme_on_device).
 *   out_cap        capacity of out in bytes
 *   rec_offsets    uint64[n+1]: record i is out[rec_offsets[i] .. rec_offsets[i+1])
 *   md             uint32[n][3]: the record's metadata fields after frame_id, zero padded to 3
 * Synchronous: returns after the records (and offsets, md) are readable by the caller.
 * RC_ERR_RECORD_TOO_LARGE / RC_ERR_OUT_TOO_SMALL leave out undefined. */
int rc_reduce_compress_batch(rc_ctx *ctx, const void *frames, uint32_t n, uint32_t first_frame_id,
                             uint8_t *out, uint64_t out_cap, uint64_t *rec_offsets, uint32_t *md);

/* Asynchronous form for device-resident pipelines: every pointer must be device memory; work is enqueued on the
 * ctx's stream and nothing is read back.  rc_ctx_sync waits for all enqueued batches and returns RC_OK, or the status
 * of the FIRST batch that failed since the previous sync (RC_ERR_RECORD_TOO_LARGE, RC_ERR_OUT_TOO_SMALL;
 * rc_last_error names the batch and frame); later batches are unaffected by an earlier failure. */
int rc_reduce_compress_batch_async(rc_ctx *ctx, const void *frames_dev, uint32_t n, uint32_t first_frame_id,
                                   uint8_t *out_dev, uint64_t out_cap, uint64_t *rec_offsets_dev, uint32_t *md_dev);
int rc_ctx_sync(rc_ctx *ctx);

/* Pipelining across batches (off by default).  A batch is two stages: the reduce kernel on the ctx's stream, then scans,
 * record layout and assembly on an internal stream, over one of two scratch sets.  Off: the ctx's stream waits for the
 * records before anything enqueued later - plain stream order.  On: it does not, so the next batch's reduce kernel
 * overlaps this batch's second stage (small latency-bound kernels that leave most of the GPU idle); a consumer of out /
 * rec_offsets / md orders itself behind the most recent batch with rc_ctx_wait_results(ctx, its_stream) (NULL = the ctx's
 * stream), or calls rc_ctx_sync.  The caller must not reuse a batch's output buffers before that: consecutive batches need their own
 * out / rec_offsets / md (two sets, alternating, are enough - batch i + 2 is ordered behind batch i; at reduction level 2 the second
 * stages of batches i and i + 1 really do run at the same time).  Batches complete in the order they were enqueued.
 * No counterpart in the reference (one frame at a time on one core, recode_writer.py:383-399). */
int rc_ctx_set_pipelined(rc_ctx *ctx, int on);
int rc_ctx_wait_results(rc_ctx *ctx, void *hip_stream);

/* ---- seam 1, host streaming form: the ingest / egress side of the writer's frame loop --------------------------------
 * Replaces the body of ReCoDeWriter.run's loop and its buffered file append (pyrecode/recode_writer.py:383-399, 605-607)
 * for callers whose frames live in host memory: RC_PIPE_SLOTS batches are in flight at once, so that the host-to-device
 * copy of batch i+1, the kernels of batch i, the device-to-host copy of batch i-1's records and the caller's file append
 * of batch i-2 overlap.  A slot owns device input / output buffers and pinned metadata; the caller owns the host buffers.
 * Call order per slot: submit -> [input_done] -> result -> fetch -> fetch_wait -> submit ...
 *
 *   rc_host_alloc / rc_host_free          page-locked host memory (for frames read from a file and for fetched records)
 *   rc_host_register / rc_host_unregister pin caller memory in place (a stack that is already in RAM)
 *   rc_pipe_submit      enqueue copy-in + all kernels of one batch; returns at once.  frames_host must stay untouched until
 *                       rc_pipe_input_done (pinned / registered memory) - pageable memory also works, the copy is then
 *                       synchronous inside the HIP runtime
 *   rc_pipe_result      wait for the batch; rec_offsets[n+1], md[n][3], *total = bytes of the n records.  Returns the batch's
 *                       status (RC_ERR_RECORD_TOO_LARGE, ...)
 *   rc_pipe_fetch       start copying the records (bytes <= *total) into dst_host; rc_pipe_fetch_wait waits for it
 *   rc_ctx_set_validation  validation frames on this path (pyrecode/recode_writer.py:402-415): for every frame of a submitted batch
 *                       whose absolute id is a multiple of `gap`, the 8-connected components of the binary map inside the region
 *                       [x0, x0 + w) x [y0, y0 + h) (<= 128 x 128: the reference's central ROI) are counted on the device, from the
 *                       frame the reduce kernel has just read.  gap 0 switches it off.
 *   rc_pipe_validation  counts[n] of the slot's batch (between rc_pipe_submit and rc_pipe_fetch_wait); 0xFFFFFFFF = not a
 *                       validation frame.  dose rate = count / (w * h), as in the reference */
#define RC_PIPE_SLOTS 3
void *rc_host_alloc(uint64_t bytes);
int rc_host_free(void *p);
int rc_host_register(void *p, uint64_t bytes);
int rc_host_unregister(void *p);
int rc_pipe_submit(rc_ctx *ctx, uint32_t slot, const void *frames_host, uint32_t n, uint32_t first_frame_id);
int rc_pipe_input_done(rc_ctx *ctx, uint32_t slot);
int rc_pipe_result(rc_ctx *ctx, uint32_t slot, uint64_t *rec_offsets, uint32_t *md, uint64_t *total);
int rc_pipe_fetch(rc_ctx *ctx, uint32_t slot, uint8_t *dst_host, uint64_t bytes);
int rc_pipe_fetch_wait(rc_ctx *ctx, uint32_t slot);
int rc_ctx_set_validation(rc_ctx *ctx, uint32_t gap, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h);
int rc_pipe_validation(rc_ctx *ctx, uint32_t slot, uint32_t *counts);

/* zstd, modelled encoder (compression_level >= 1): the ctx fits its entropy tables to a sample of the FIRST batch it sees and
 * keeps them (every frame carries the tables it was coded with, so any model is valid for any data - only the ratio suffers when
 * the data drifts away from the sample).  rc_ctx_refit_model makes the NEXT batch the sample again (one synchronous step), e.g.
 * after a change of dose or detector mode; a no-op for other schemes.  No counterpart in the reference (libzstd adapts per call). */
int rc_ctx_refit_model(rc_ctx *ctx);

/* Packed binary map (ceil(nx*ny/8) bytes, LSB-first) of frame i of the most recent batch: what the third element
 * of _reduce_compress's return value carries for validation frames (recode_writer.py:386,402-415,557). */
int rc_get_binary_map(rc_ctx *ctx, uint32_t i, uint8_t *bitmap_out);
/* The reference only consumes binary_frame for validation frames (recode_writer.py:402-415).  With a device codec the
 * raw maps are an extra HBM write; a caller that never asks for them can switch that off (default: kept).  Without a
 * device codec the raw map is part of the record and always produced. */
int rc_ctx_keep_binary_maps(rc_ctx *ctx, int on);

/* Per-stage device time of the most recent batch in milliseconds (HIP events), keyed like the reference's
 * run metrics (recode_writer.py:451,457,479,506,512,555):
 *   [0] frame_thresholding_and_counting_time + frame_binary_image_packing_time (fused reduce kernel; with LZ4 the
 *       bitmap compression is fused in here as well)
 *   [1] frame_binary_image_compression_time when the codec is a kernel of its own (zstd)   [2] per-frame scans
 *   [3] record layout + frame_pixel_intensity_packing_time + frame_pixel_intensity_compression_time (assemble kernel)
 *   [4] whole batch (frame_time * n)
 * Only filled by the synchronous entry point. */
int rc_get_stage_ms(rc_ctx *ctx, float ms[5]);

/* Stage timing of the ASYNCHRONOUS path: with profiling on, every rc_reduce_compress_batch_async brackets its stages
 * with HIP events on the ctx's stream; rc_ctx_sync folds them into running sums (same 5 slots as rc_get_stage_ms).
 * rc_ctx_set_profiling also clears the sums.  Used by bench.py to measure the dominant kernel inside the timed region.
 * on = k > 1: events around every k-th batch only, starting with the next one (a timing event is a packet of its own on the
 * stream, a few microseconds between two reduce kernels; rc_ctx_get_profile's `batches` counts the bracketed ones). */
int rc_ctx_set_profiling(rc_ctx *ctx, int on);
int rc_ctx_get_profile(rc_ctx *ctx, double sum_ms[5], uint64_t *batches);

/* ---- seam 2: compressor backend -----------------------------------------------------------------------
 * compress()/de_compress() of pyrecode/recode_compressors.py:82-120 / :40-79 for the device codecs.
 * Host or device pointers.  *out_n receives the produced byte count; rc_decompress also sets it to the required size
 * when it returns RC_ERR_OUT_TOO_SMALL (dst may then be NULL with dst_cap 0: a size query).
 * Every scheme rc_scheme_on_device() reports (LZ4, zstd, blosc-lz4) is both encoded and decoded here.  Decoding covers
 * what this library writes and the stock encoders' frames of the same kind: LZ4 frames (independent or linked blocks),
 * blosc1-LZ4 chunks; zstd frames inside the device decoder's subset (rc_zstd_dec.h: single-stream Huffman or raw literals,
 * predefined / described / repeated sequence tables, repeat-offset matches) - for a zstd frame outside it (a stock
 * encoder's 4-stream literals, real offsets) rc_decompress returns RC_ERR_UNSUPPORTED before doing any work and the caller
 * uses the stock decoder (pyrecode_amd/recode_compressors.py::de_compress does).
 * Runs on the caller's current GPU, or on `RC_DEVICE` (env) when that is set. */
int rc_compress(uint32_t scheme, uint32_t level, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap,
                uint64_t *out_n);
int rc_decompress(uint32_t scheme, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *out_n);
uint64_t rc_compress_bound(uint32_t scheme, uint64_t n);

/* ---- seam 3: the native c_recode.Reader methods (pyrecode/pyrecode.cpp:143-150) ---------------------------
 * get_frame_sparse -> _unpack_frame_sparse (pyrecode.cpp:95-119, c_extensions/reader.h:10-68): for every set bit of
 * the bitmap in row-major order write (row, col, val) as three uint64; level 1: val = next d-bit LSB-first field
 * of pixvals; other levels: val = 1.  out must hold 3 * popcount(bitmap) uint64 (the reference sizes it
 * nx*ny*3, recode_reader.py:111-115); out_cap_triplets bounds it.  Returns nnz >= 0 or a negative rc_status.
 * out == NULL with out_cap_triplets == 0 is a counting call: returns popcount(bitmap) and writes nothing. */
int64_t rc_unpack_frame_sparse(uint32_t nx, uint32_t ny, uint32_t bit_depth, const uint8_t *bitmap,
                               const uint8_t *pixvals, uint64_t pixvals_bytes, uint64_t *out,
                               uint64_t out_cap_triplets, uint32_t reduction_level);
/* Batched form of the reader's per-frame work: n stored frames -> decompress both streams -> sparse expand, in one call
 * with no host round trip in between.  Replaces, for n frames at once, ReCoDeReader._get_frame_sparse
 * (pyrecode/recode_reader.py:379-471: de_compress on the binary-map stream and on the value stream,
 * recode_compressors.py:40-79, then c_recode get_frame_sparse, pyrecode.cpp:95-119).
 *   nx, ny, bit_depth, reduction_level (1 or 3), op_mode, scheme    header fields of the file
 *   data          host or device memory: the n frames' data blobs back to back, as they lie in a merged file (per frame: the
 *                 binary-map stream, then the value stream).  Host memory is copied in while the streams' block headers are
 *                 walked (page-locked memory - rc_host_alloc - makes that copy asynchronous); device memory is decoded where it lies
 *   sizes         uint32[n][3]: bytes of the binary-map stream, bytes of the value stream, bytes of the DEcompressed value
 *                 stream (the rows of the file's metadata table; mode 0: {nb, n_packed, n_packed})
 *   nnz_prefix    uint64[n+1] out: exclusive prefix of the frames' set-pixel counts (frame i's triplets are
 *                 [nnz_prefix[i], nnz_prefix[i+1]))
 *   triplets      uint64[cap][3] out (host or device): (row, col, value) in row-major order per frame, frames in order;
 *                 may be NULL with cap 0 (a counting call)
 * Device decoders: mode 0 (stored pieces), LZ4 frames with independent blocks, zstd frames inside the subset this library
 * writes (rc_zstd_dec.h).  Anything else returns RC_ERR_UNSUPPORTED before any work is done and the caller falls back to
 * the per-frame path with the stock decoder; a stream that is inside the subset on its face but does not decode to the frame's
 * shape (a foreign encoder's larger blocks, or damage) returns RC_ERR_CORRUPT - callers that can fall back should do so for both
 * codes and let the stock decoder judge.  RC_ERR_OUT_TOO_SMALL: nnz_prefix is valid, triplets untouched.  At reduction level 1
 * sum(8 * sizes[i][2] / bit_depth) bounds the count, so one call with that capacity does (ReCoDeReader.get_frames_triplets).
 * The host walk runs on a small pool of worker threads that stays with the process. */
int rc_expand_frames(uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t reduction_level, uint32_t op_mode, uint32_t scheme,
                     const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *nnz_prefix, uint64_t *triplets, uint64_t cap);
/* The same work in two halves, for a reader that streams through a file: _submit queues one batch (host walk of the block headers,
 * copy-in, decoders, count, emit) on one of two slots and returns; _wait(slot) blocks until that batch is done and reports like
 * rc_expand_frames.  With two batches in flight the host walk and copy-in of one run while the device decodes the other.
 *   slot            0 or 1; a slot holds one batch at a time (_submit on a slot with a batch waiting: RC_ERR_BAD_ARG).  The synchronous
 *                   rc_expand_frames owns resources of its own, so it may be called (e.g. to redo ONE batch) while batches are queued
 *   triplets_dev    device memory, or page-locked host memory (rc_host_alloc: staged on the device, then ONE asynchronous copy of all
 *                   cap entries - keep cap tight; contents unspecified when _wait reports an error), cap entries
 *                   (level 1: sum(8 * sizes[i][2] / bit_depth) bounds the count)
 *   data            must stay valid until _wait(slot) returns; sizes is consumed by _submit
 * Streams outside the device decoders' subset are reported by _submit (RC_ERR_UNSUPPORTED / RC_ERR_CORRUPT, nothing left pending);
 * what only the device can see (a block that does not decode to its size, cap too small) by _wait. */
int rc_expand_frames_submit(uint32_t slot, uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t reduction_level, uint32_t op_mode,
                            uint32_t scheme, const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *triplets_dev, uint64_t cap);
int rc_expand_frames_wait(uint32_t slot, uint64_t *nnz_prefix);
/* The same two calls with the output laid out as what the reference's reader makes of the triplets - the three arrays of a scipy COO
 * matrix (recode_reader.py:466-469): `coo` = int32 rows[cap] | int32 columns[cap] | uint16 values[cap] (10 * cap bytes; entry i of the
 * batch at index i of each array, frame f's entries at nnz_prefix[f] .. nnz_prefix[f+1]).  10 instead of 24 bytes per set pixel cross
 * the link, and the host has nothing to split.  bit_depth <= 16.  rc_expand_frames_coo_submit is waited for with
 * rc_expand_frames_wait like a triplet batch. */
int rc_expand_frames_coo(uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t reduction_level, uint32_t op_mode, uint32_t scheme,
                         const uint8_t *data, const uint32_t *sizes, uint32_t n, uint64_t *nnz_prefix, void *coo, uint64_t cap);
int rc_expand_frames_coo_submit(uint32_t slot, uint32_t nx, uint32_t ny, uint32_t bit_depth, uint32_t reduction_level, uint32_t op_mode,
                                uint32_t scheme, const uint8_t *data, const uint32_t *sizes, uint32_t n, void *coo_dev, uint64_t cap);
/* The host half of a batch whose streams only a STOCK decoder takes - files the reference's writer produced with zstandard /
 * lz4.frame / zlib (recode_compressors.py:82-120): linked 64 KiB LZ4 blocks, 4-stream Huffman literals and real offsets in zstd,
 * deflate.  Each is one serial chain, which the reference walks with one library call per stream (recode_compressors.py:40-79:
 * zlib.decompress :43, zstd :46, lz4.frame.decompress :49).  rc_host_decode_streams makes the SAME library's call (libzstd.so.1 /
 * liblz4.so.1 / libz.so.1, bound at run time) for n streams at once on worker threads: stream i is src[spans[4i] .. + spans[4i+1])
 * and must decode to exactly spans[4i+3] bytes at dst + spans[4i+2] (host pointers; dst is typically the page-locked stored-pieces
 * image that rc_expand_frames / _submit then takes with op_mode 0).  threads 0 = min(16, cores).  Schemes 1 (zstd), 2 (LZ4), 0 (zlib).
 * RC_ERR_UNSUPPORTED: no such library on this host (rc_host_decoder_available says so beforehand);
 * RC_ERR_CORRUPT: the stock decoder rejected a stream or it decodes to another size.  No GPU is touched. */
int rc_host_decoder_available(uint32_t scheme);     /* 1 / 0 */
int rc_host_decode_streams(uint32_t scheme, const uint8_t *src, uint8_t *dst, const uint64_t *spans, uint32_t n, uint32_t threads);
/* The last step of the reference's get_frame (recode_reader.py:466-469: coo_matrix((data, (row, col)))) for a frame's triplets: its
 * uint64 (row, col, value) rows -> int32 rows, int32 columns and values as unsigned integers of val_bytes (1 / 2 / 4 / 8) bytes, in ONE
 * pass.  Host pointers; no GPU is touched. */
int rc_split_triplets(const uint64_t *triplets, uint64_t n, int32_t *row, int32_t *col, void *val, uint32_t val_bytes);
/* bit_pack_pixel_intensities -> _bit_pack_pixel_intensities (reader.h:105-140) with the intended semantics of
 * the numba _bit_pack (recode_writer.py:637-652): zero, then LSB-first d-bit fields.  out_n = ceil(n*d/8). */
int rc_bit_pack(const uint16_t *pixvals, uint64_t n, uint32_t bit_depth, uint8_t *out, uint64_t out_n);
/* bit_unpack_pixel_intensities -> intended semantics of reader.h:74-99: n d-bit fields -> uint64[n]. */
int rc_bit_unpack(const uint8_t *packed, uint64_t packed_bytes, uint64_t n, uint32_t bit_depth, uint64_t *out);

/* ---- synthetic stacks for tests / bench --------------------------------------------------
 * Counter-based integer generator, identical on host (pyrecode_amd/synth.py) and device:
 * dark in [80,120]; Bernoulli(sparsity_ppm / 1e6) events of amplitude [1,2047] above dark; background <= dark. */
int rc_synth_dark(int device_id, uint32_t seed, uint64_t n_pixels, uint16_t *dark_dev);
int rc_synth_frames(int device_id, uint32_t seed, uint32_t first_frame, uint32_t n_frames, uint64_t n_pixels,
                    uint32_t sparsity_ppm, const uint16_t *dark_dev, uint16_t *frames_dev);
/* Detector-like variant: events come in clusters of 1..6 pixels (mean 4.1) inside a 2 x 3 window anchored at a seed pixel;
 * seed_ppm = seeds per million pixels (10 700 gives ~4.3 % set pixels: the real acquisition the reference's notebook records,
 * examples/Reading_ReCoDe_v0.1_Files.ipynb cells 7 / 17).  Same amplitudes and background as rc_synth_frames. */
int rc_synth_frames_clustered(int device_id, uint32_t seed, uint32_t first_frame, uint32_t n_frames, uint32_t nx, uint32_t ny,
                              uint32_t seed_ppm, const uint16_t *dark_dev, uint16_t *frames_dev);

#ifdef __cplusplus
}
#endif
#endif /* RECODE_HIP_H */
