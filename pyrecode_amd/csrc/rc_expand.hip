// rc_expand.hip - reader-side kernels: sparse expand (bitmap + packed pixvals -> (row, col, val) triplets), stand-alone
// d-bit pack / unpack, and the synthetic stack generator (gfx950).
//
//   A10  _unpack_frame_sparse            pyrecode/c_extensions/reader.h:10-68  (via pyrecode.cpp:95-119)
//   A5   _bit_pack_pixel_intensities     pyrecode/c_extensions/reader.h:105-140 (intended semantics = recode_writer.py:637-652)
//        _bit_unpack_pixel_intensities   pyrecode/c_extensions/reader.h:74-99   (intended semantics)
#include "rc_expand.h"

namespace rc {


// bitmap is padded with zero bytes to a multiple of 8 by the host wrapper, so 8-byte loads are always in bounds.
// 64 bitmap bits of word i with the bits of pixels at or past N cleared: a well-formed bitmap has them zero, a foreign or
// damaged file may not, and the count pass and the emit pass must agree on what they see
__device__ __forceinline__ uint64_t expand_word(const uint8_t *__restrict__ bitmap, uint64_t nb8, uint64_t N, uint64_t i)
{
    if (i >= nb8) return 0;
    const u32x2 v = reinterpret_cast<const u32x2 *>(bitmap)[i];
    uint64_t bits = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    const uint64_t k0 = i * 64;
    if (k0 + 64 > N) bits = k0 >= N ? 0 : bits & ((1ull << (N - k0)) - 1);
    return bits;
}

__global__ __launch_bounds__(WG) void k_expand_count(const uint8_t *__restrict__ bitmap, uint64_t nb8, uint64_t N,
                                                       uint32_t *__restrict__ blk_cnt)
{
    __shared__ uint32_t sm[WAVES + 1];
    const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;  // 8-byte word index
    const uint32_t c = (uint32_t)__builtin_popcountll(expand_word(bitmap, nb8, N, i));
    uint32_t tot;
    (void)block_excl_scan(c, sm, &tot);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of blk_cnt[0..nblk) in place into blk_off, total -> *nnz
__global__ __launch_bounds__(WG) void k_expand_scan(const uint32_t *__restrict__ blk_cnt, uint32_t *__restrict__ blk_off,
                                                      uint32_t nblk, uint64_t *__restrict__ nnz)
{
    __shared__ uint32_t sm[WAVES + 1];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += WG) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < nblk ? blk_cnt[b] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, sm, &tot);
        if (b < nblk) blk_off[b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *nnz = carry;
}

// d-bit LSB-first field number `idx` of a packed stream of `nbytes` bytes (bytes past the end read as zero)
__device__ __forceinline__ uint64_t read_field(const uint8_t *__restrict__ p, uint64_t nbytes, uint64_t idx, uint32_t d)
{
    const uint64_t bit = idx * d;
    uint64_t byte = bit >> 3;
    const uint32_t sh = (uint32_t)(bit & 7);
    const uint32_t need = (sh + d + 7) >> 3;  // <= 9 bytes for d <= 64
    uint64_t lo = 0, hi = 0;
    for (uint32_t k = 0; k < need && k < 8; ++k)
        if (byte + k < nbytes) lo |= (uint64_t)p[byte + k] << (8 * k);
    if (need > 8 && byte + 8 < nbytes) hi = p[byte + 8];
    uint64_t v = lo >> sh;
    if (sh && need > 8) v |= hi << (64 - sh);
    return d >= 64 ? v : (v & ((1ull << d) - 1));
}

__global__ __launch_bounds__(WG) void k_expand_emit(const uint8_t *__restrict__ bitmap, uint64_t nb8, uint64_t N, uint32_t nx,
                                                      const uint32_t *__restrict__ blk_off,
                                                      const uint8_t *__restrict__ pix, uint64_t pix_bytes, uint32_t d,
                                                      uint32_t level, uint64_t cap, uint64_t *__restrict__ out)
{
    __shared__ uint32_t sm[WAVES + 1];
    const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t bits = expand_word(bitmap, nb8, N, i);
    const uint64_t k0 = i * 64;
    uint32_t tot;
    uint64_t rank = blk_off[blockIdx.x] + block_excl_scan((uint32_t)__builtin_popcountll(bits), sm, &tot);
    for (; bits; bits &= bits - 1, ++rank) {
        if (rank >= cap) break;
        const uint64_t k = k0 + (uint64_t)__builtin_ctzll(bits);
        const uint32_t row = (uint32_t)(k / nx), col = (uint32_t)(k - (uint64_t)row * nx);
        out[3 * rank] = row;
        out[3 * rank + 1] = col;
        out[3 * rank + 2] = level == 1 ? read_field(pix, pix_bytes, rank, d) : 1ull;
    }
}

void launch_expand_count(const uint8_t *bitmap_pad8, uint64_t nb8, uint64_t N, uint32_t *blk_cnt, uint32_t *blk_off,
                         uint64_t *nnz_dev, hipStream_t s)
{
    const uint32_t nblk = (uint32_t)((nb8 + WG - 1) / WG);
    hipLaunchKernelGGL(k_expand_count, dim3(nblk), dim3(WG), 0, s, bitmap_pad8, nb8, N, blk_cnt);
    hipLaunchKernelGGL(k_expand_scan, dim3(1), dim3(WG), 0, s, blk_cnt, blk_off, nblk, nnz_dev);
}
void launch_expand_emit(const uint8_t *bitmap_pad8, uint64_t nb8, uint64_t N, uint32_t nx, const uint32_t *blk_off,
                        const uint8_t *pix, uint64_t pix_bytes, uint32_t d, uint32_t level, uint64_t cap, uint64_t *out,
                        hipStream_t s)
{
    const uint32_t nblk = (uint32_t)((nb8 + WG - 1) / WG);
    hipLaunchKernelGGL(k_expand_emit, dim3(nblk), dim3(WG), 0, s, bitmap_pad8, nb8, N, nx, blk_off, pix, pix_bytes, d, level,
                       cap, out);
}

// ---- batched form (rc_expand_frames): n frames' decoded bitmaps / value streams, frame f at bm + f * bm_stride ------------------
__global__ __launch_bounds__(WG) void k_expand_count_b(const uint8_t *__restrict__ bm, uint64_t bm_stride, uint64_t nb8, uint64_t N,
                                                         uint32_t nblk, uint32_t *__restrict__ blk_cnt)
{
    __shared__ uint32_t sm[WAVES + 1];
    const uint32_t f = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    const uint32_t c = (uint32_t)__builtin_popcountll(expand_word(bm + f * bm_stride, nb8, N, i));
    uint32_t tot;
    (void)block_excl_scan(c, sm, &tot);
    if (threadIdx.x == 0) blk_cnt[(uint64_t)f * nblk + blockIdx.x] = tot;
}
// one workgroup per frame: blk_cnt row -> exclusive prefix (in place into blk_off), frame_nnz[f]
__global__ __launch_bounds__(WG) void k_expand_scan_b(const uint32_t *__restrict__ blk_cnt, uint32_t *__restrict__ blk_off, uint32_t nblk,
                                                        uint64_t *__restrict__ frame_nnz)
{
    __shared__ uint32_t sm[WAVES + 1];
    const uint32_t f = blockIdx.x;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += WG) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < nblk ? blk_cnt[(uint64_t)f * nblk + b] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, sm, &tot);
        if (b < nblk) blk_off[(uint64_t)f * nblk + b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) frame_nnz[f] = carry;
}
// single workgroup: frame_nnz[0..n) -> frame_base[0..n] (exclusive prefix, total at [n]).  With err != NULL it is also the gate of the
// emit kernel that may already be queued behind it: a total above cap sets bit 1 of *err, a frame whose value stream is shorter
// than its set pixels need (level 1: pv_bytes) bit 2; k_expand_emit_b writes nothing once *err is set.
__global__ __launch_bounds__(WG) void k_expand_bases(const uint64_t *__restrict__ frame_nnz, uint64_t *__restrict__ frame_base, uint32_t n,
                                                       const uint32_t *__restrict__ pv_bytes, uint32_t d, uint32_t level, uint64_t cap,
                                                       int *__restrict__ err)
{
    __shared__ uint64_t part[WG];
    const uint32_t per = (n + WG - 1) / WG, lo = threadIdx.x * per, hi = min(lo + per, n);
    uint64_t s = 0;
    for (uint32_t f = lo; f < hi; ++f) {
        const uint64_t c = frame_nnz[f];
        s += c;
        if (err && level == 1 && (c * d + 7) / 8 > pv_bytes[f]) atomicOr(err, 4);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t i = 0; i < threadIdx.x; ++i) base += part[i];
    for (uint32_t f = lo; f < hi; ++f) { frame_base[f] = base; base += frame_nnz[f]; }
    if (threadIdx.x == WG - 1) {
        frame_base[n] = base;
        if (err && base > cap) atomicOr(err, 2);
    }
}
// A workgroup's triplets are ONE contiguous range of the output (frame base + block offset, in row-major order), so they are
// assembled in LDS and leave as consecutive 8-byte stores - a lane writing its three words itself spreads every store instruction
// over a dozen partially written lines (181 us for 64 frames of 4096 x 4096 at 1 %; this form: see DESIGN.md).  Blocks with more set
// pixels than the stage holds write directly.  Row / column of a word's first pixel come from ONE division (frames have fewer than
// 2^32 pixels: nx, ny <= 65 535), the pixels behind it step the column.
constexpr uint32_t EMIT_STAGE = 1024;   // triplets
// COO = false: out = uint64_t[cap][3], the reference's (row, col, value) rows (pyrecode.cpp:95-119).
// COO = true: out = int32 rows[cap] | int32 columns[cap] | uint16 values[cap] - the three arrays of the scipy COO matrix the reference's
// reader wraps the rows into (recode_reader.py:466-469), 10 instead of 24 bytes per set pixel on the link and no split on the host.
template <bool COO>
__global__ __launch_bounds__(WG) void k_expand_emit_b(const uint8_t *__restrict__ bm, uint64_t bm_stride, uint64_t nb8, uint64_t N, uint32_t nx,
                                                        uint32_t nblk, const uint32_t *__restrict__ blk_off, const uint64_t *__restrict__ frame_base,
                                                        const uint8_t *__restrict__ pv, uint64_t pv_stride, const uint32_t *__restrict__ pv_bytes,
                                                        uint32_t d, uint32_t level, uint64_t cap, void *__restrict__ out_any,
                                                        const int *__restrict__ err)
{
    __shared__ uint32_t sm[WAVES + 1];
    __shared__ uint64_t s_trip[COO ? 1 : 3 * EMIT_STAGE];
    __shared__ uint32_t s_row[COO ? EMIT_STAGE : 1], s_col[COO ? EMIT_STAGE : 1];
    __shared__ uint16_t s_val[COO ? EMIT_STAGE : 1];
    if (err && *err) return;   // (workgroup-uniform: k_expand_bases, a decoder or nobody has set it before this kernel started)
    uint64_t *out = static_cast<uint64_t *>(out_any);
    int32_t *o_row = static_cast<int32_t *>(out_any), *o_col = o_row + cap;
    uint16_t *o_val = reinterpret_cast<uint16_t *>(o_col + cap);
    const uint32_t f = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t bits = expand_word(bm + f * bm_stride, nb8, N, i);
    const uint64_t k0 = i * 64;
    uint32_t tot;
    const uint32_t local = block_excl_scan((uint32_t)__builtin_popcountll(bits), sm, &tot);
    if (tot == 0) return;
    const uint64_t wg_rank = blk_off[(uint64_t)f * nblk + blockIdx.x];      // rank of the workgroup's first set pixel inside the frame
    const uint64_t base = frame_base[f];
    const uint8_t *pix = pv + f * pv_stride;
    const uint64_t pix_bytes = level == 1 ? pv_bytes[f] : 0;
    const bool staged = tot <= EMIT_STAGE && base + wg_rank + tot <= cap;    // (uniform; a range that would cross cap keeps the per-entry check)
    if (bits) {
        uint32_t row = (uint32_t)((uint32_t)k0 / nx), col = (uint32_t)k0 - row * nx, prev = 0;
        uint32_t r = local;
        for (; bits; bits &= bits - 1, ++r) {
            const uint32_t b = (uint32_t)__builtin_ctzll(bits);
            col += b - prev;
            prev = b;
            while (col >= nx) { col -= nx; ++row; }
            const uint64_t rank = wg_rank + r;
            const uint64_t val = level != 1 ? 1ull : ((d == 16 && 2 * rank + 2 <= pix_bytes) ? (uint64_t)reinterpret_cast<const uint16_t *>(pix)[rank]   // (value streams are 16-byte aligned)
                                                                                                : read_field(pix, pix_bytes, rank, d));
            if (staged) {
                if (COO) { s_row[r] = row; s_col[r] = col; s_val[r] = (uint16_t)val; }
                else { s_trip[3 * r] = row; s_trip[3 * r + 1] = col; s_trip[3 * r + 2] = val; }
            } else {
                if (base + rank >= cap) break;
                if (COO) { o_row[base + rank] = (int32_t)row; o_col[base + rank] = (int32_t)col; o_val[base + rank] = (uint16_t)val; }
                else {
                    uint64_t *o = out + 3 * (base + rank);
                    o[0] = row; o[1] = col; o[2] = val;
                }
            }
        }
    }
    if (!staged) return;
    __syncthreads();
    if (COO) {
        const uint64_t at = base + wg_rank;
        for (uint32_t j = threadIdx.x; j < tot; j += WG) { o_row[at + j] = (int32_t)s_row[j]; o_col[at + j] = (int32_t)s_col[j]; o_val[at + j] = s_val[j]; }
    } else {
        uint64_t *o = out + 3 * (base + wg_rank);
        for (uint32_t j = threadIdx.x; j < 3 * tot; j += WG) o[j] = s_trip[j];
    }
}
void launch_expand_batch_count(const uint8_t *bm, uint64_t bm_stride, uint64_t nb8, uint64_t N, uint32_t n, uint32_t *blk_cnt, uint32_t *blk_off,
                               uint64_t *frame_nnz, uint64_t *frame_base, hipStream_t s, const uint32_t *pv_bytes, uint32_t d, uint32_t level,
                               uint64_t cap, int *err)
{
    const uint32_t nblk = (uint32_t)((nb8 + WG - 1) / WG);
    hipLaunchKernelGGL(k_expand_count_b, dim3(nblk, n), dim3(WG), 0, s, bm, bm_stride, nb8, N, nblk, blk_cnt);
    hipLaunchKernelGGL(k_expand_scan_b, dim3(n), dim3(WG), 0, s, blk_cnt, blk_off, nblk, frame_nnz);
    hipLaunchKernelGGL(k_expand_bases, dim3(1), dim3(WG), 0, s, frame_nnz, frame_base, n, pv_bytes, d, level, cap, err);
}
void launch_expand_batch_emit(const uint8_t *bm, uint64_t bm_stride, uint64_t nb8, uint64_t N, uint32_t nx, uint32_t n, const uint32_t *blk_off,
                              const uint64_t *frame_base, const uint8_t *pv, uint64_t pv_stride, const uint32_t *pv_bytes, uint32_t d,
                              uint32_t level, uint64_t cap, void *out, hipStream_t s, const int *err, bool coo)
{
    const uint32_t nblk = (uint32_t)((nb8 + WG - 1) / WG);
    if (coo)
        hipLaunchKernelGGL(k_expand_emit_b<true>, dim3(nblk, n), dim3(WG), 0, s, bm, bm_stride, nb8, N, nx, nblk, blk_off, frame_base, pv, pv_stride,
                           pv_bytes, d, level, cap, out, err);
    else
        hipLaunchKernelGGL(k_expand_emit_b<false>, dim3(nblk, n), dim3(WG), 0, s, bm, bm_stride, nb8, N, nx, nblk, blk_off, frame_base, pv, pv_stride,
                           pv_bytes, d, level, cap, out, err);
}

// ---- stand-alone pack / unpack ------------------------------------------------------------------------------
__global__ void k_bit_pack(const uint16_t *__restrict__ vals, uint64_t n, uint32_t d, uint8_t *__restrict__ out, uint64_t out_n)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= out_n) return;
    const uint32_t dmask = d >= 16 ? 0xFFFFu : ((1u << d) - 1);
    const uint64_t bit0 = b * 8;
    uint64_t v = bit0 / d;
    const uint32_t o = (uint32_t)(bit0 - v * d);
    uint32_t acc = v < n ? ((vals[v] & dmask) >> o) : 0;
    uint32_t filled = d - o;
    while (filled < 8) {
        ++v;
        if (v < n) acc |= (uint32_t)(vals[v] & dmask) << filled;
        filled += d;
    }
    out[b] = (uint8_t)acc;
}
__global__ void k_bit_unpack(const uint8_t *__restrict__ packed, uint64_t nbytes, uint64_t n, uint32_t d, uint64_t *__restrict__ out)
{
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) out[v] = read_field(packed, nbytes, v, d);
}
void launch_bit_pack(const uint16_t *vals, uint64_t n, uint32_t d, uint8_t *out, uint64_t out_n, hipStream_t s)
{
    if (out_n == 0) return;
    hipLaunchKernelGGL(k_bit_pack, dim3((uint32_t)((out_n + 255) / 256)), dim3(256), 0, s, vals, n, d, out, out_n);
}
void launch_bit_unpack(const uint8_t *packed, uint64_t nbytes, uint64_t n, uint32_t d, uint64_t *out, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_bit_unpack, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, packed, nbytes, n, d, out);
}

// ---- synthetic stacks, integer counter-based generator mirrored in pyrecode_amd/synth.py -------------
__host__ __device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__global__ void k_synth_dark(uint32_t seed, uint64_t N, uint16_t *__restrict__ dark)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < N; i += stride) dark[i] = (uint16_t)(80u + mix32((uint32_t)i ^ mix32(seed ^ 0xD1B54A32u)) % 41u);
}
__global__ void k_synth_frames(uint32_t seed, uint32_t first_frame, uint32_t nframes, uint64_t N, uint32_t thresh24,
                               const uint16_t *__restrict__ dark, uint16_t *__restrict__ frames)
{
    const uint32_t z = blockIdx.y;
    const uint32_t fkey = mix32(seed + 0x9E3779B9u * (first_frame + z + 1u));
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint16_t *fr = frames + (uint64_t)z * N;
    for (; i < N; i += stride) {
        const uint32_t h = mix32((uint32_t)i ^ fkey);
        const uint32_t h2 = mix32(h ^ 0x68E31DA4u);
        const uint32_t dk = dark[i];
        const uint32_t v = (h & 0xFFFFFFu) < thresh24 ? dk + 1u + (h2 % 2047u) : (h2 % (dk + 1u));
        fr[i] = (uint16_t)v;
    }
    (void)nframes;
}
// Detector-like events (the one real-data anchor the reference records: 4096^2, 12 bit, ~4.3 % of the pixels set, in small
// clusters - examples/Reading_ReCoDe_v0.1_Files.ipynb): pixel q is a SEED with probability thresh24 / 2^24; a seed lights itself
// and each of the other five cells of the 2 x 3 window it anchors (top-left) with probability 5/8 (3 hash bits per cell), i.e.
// clusters of 1..6 pixels, 4.1 on average.  A pixel is an event iff one of the six windows covering it lights it.
__global__ void k_synth_frames_clustered(uint32_t seed, uint32_t first_frame, uint32_t nx, uint32_t ny, uint32_t thresh24,
                                         const uint16_t *__restrict__ dark, uint16_t *__restrict__ frames)
{
    const uint32_t z = blockIdx.y;
    const uint32_t fkey = mix32(seed + 0x9E3779B9u * (first_frame + z + 1u));
    const uint64_t N = (uint64_t)nx * ny;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint16_t *fr = frames + (uint64_t)z * N;
    for (; i < N; i += stride) {
        const uint32_t y = (uint32_t)(i / nx), x = (uint32_t)(i - (uint64_t)y * nx);
        bool ev = false;
        for (uint32_t dy = 0; dy < 2 && dy <= y; ++dy)
            for (uint32_t dx = 0; dx < 3 && dx <= x; ++dx) {
                const uint32_t q = (uint32_t)(i - (uint64_t)dy * nx - dx);
                const uint32_t hq = mix32(q ^ fkey);
                if ((hq & 0xFFFFFFu) >= thresh24) continue;
                const uint32_t cell = dy * 3 + dx;
                ev |= cell == 0 || ((mix32(hq ^ 0x3C6EF372u) >> (3 * cell)) & 7u) < 5u;
            }
        const uint32_t h2 = mix32(mix32((uint32_t)i ^ fkey) ^ 0x68E31DA4u);
        const uint32_t dk = dark[i];
        fr[i] = (uint16_t)(ev ? dk + 1u + (h2 % 2047u) : (h2 % (dk + 1u)));
    }
}
void launch_synth_frames_clustered(uint32_t seed, uint32_t first_frame, uint32_t nframes, uint32_t nx, uint32_t ny, uint32_t seed_ppm,
                                   const uint16_t *dark, uint16_t *frames, hipStream_t s)
{
    const uint32_t thresh24 = (uint32_t)(((uint64_t)seed_ppm << 24) / 1000000ull);
    hipLaunchKernelGGL(k_synth_frames_clustered, dim3(1024, nframes), dim3(256), 0, s, seed, first_frame, nx, ny, thresh24, dark, frames);
}
void launch_synth_dark(uint32_t seed, uint64_t N, uint16_t *dark, hipStream_t s)
{
    hipLaunchKernelGGL(k_synth_dark, dim3(2048), dim3(256), 0, s, seed, N, dark);
}
void launch_synth_frames(uint32_t seed, uint32_t first_frame, uint32_t nframes, uint64_t N, uint32_t sparsity_ppm,
                         const uint16_t *dark, uint16_t *frames, hipStream_t s)
{
    const uint32_t thresh24 = (uint32_t)(((uint64_t)sparsity_ppm << 24) / 1000000ull);
    hipLaunchKernelGGL(k_synth_frames, dim3(1024, nframes), dim3(256), 0, s, seed, first_frame, nframes, N, thresh24, dark,
                       frames);
}

}  // namespace rc
