// rc_l2.hip - reduction level 2: one summary statistic per connected component of the binary map.
//
// Reference intent (the reference's own L2 path cannot run): pyrecode/recode_writer.py:443-446
//   labeled, n = scipy.ndimage.label(binary_frame, structure=ones((3,3)))      8-connectivity, labels in raster order of each
//   stats = get_summary_stats_nb(labeled, frame, 0, dtype, 'max' | 'sum')      component's first pixel; statistic of the RAW
//   (pyrecode/utils/converters.py:262-297)                                     frame values, cast to the source dtype
// The record then carries the full binary map and the statistics where L1 carries the residuals (recode_writer.py:461-525).
//
// Device formulation (round 5).  The reduce kernel leaves, per tile of 4096 pixels, the raw binary map (64 words of 64 pixels) and the
// raw values of its set pixels in raster order (the tile's value slot).  A set pixel's ID inside its frame is tile * 4096 + its rank
// among the tile's set pixels: ids grow in raster order, need no per-frame prefix (no scan in front of this stage) and address two
// sparse arrays, parent[] and stat[], of which only the entries of set pixels are ever touched.  Four passes, a wavefront per tile:
//   k_l2_prep    a lane per word: popcount, wave scan -> the word's rank base inside its tile (u16 directory); parent[id] = id and
//                stat[id] = 0 for the tile's set pixels (rank-parallel, whole lines)
//   k_l2_link    a lane per word: which of its pixels have a set neighbour among W, NW, N, NE comes from WORD arithmetic on the word, its
//                left neighbour and the three words around it one row up (funnel-shifted: the row length need not be a multiple of
//                64) - at 1 % Bernoulli 96 % of the set pixels have none and are done.  The others union with N, or with W / NW and
//                NE (the rest of the four are connected to those through their own links): neighbour id = directory entry + a
//                popcount; union-find with the smaller id as root (atomicMin links, path halving)
//   k_l2_stats   a lane per set pixel: a pixel that is not its own root finds it and adds its raw value (atomicMax / atomicAdd)
//   k_l2_emit    a lane per set pixel: roots - in id order, i.e. in scipy's label order - leave their statistic (their own value
//                joined with what the others added) as the tile's NEW value list: compacted in LDS, bit-packed to the tile-local d-bit
//                stream exactly as the reduce kernel leaves level-1 residuals, written back to the tile's slot with the tile's new
//                count.  From there on the batch IS a level-1 batch: scans, record layout, k_gather and every codec run unchanged.
// Before round 5 the stage compacted every set pixel into batch-global arrays (position, value, parent, accumulator: 14 bytes each), looked
// every pixel's four neighbours up in the map one by one, and emitted per frame with ONE workgroup: 870 us per 64 frames of 4096^2 at 1 %
// (k_l2_union 281, k_l2_index 216, k_l2_emit 161, k_l2_stats 99; profiles/r05_exp4), 1.5 ms on clustered events.
#include "rc_launch.h"
#include "rc_pack.h"

namespace rc {

__device__ __forceinline__ uint32_t uf_find(uint32_t *__restrict__ parent, uint32_t x)
{
    uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const uint32_t gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_union(uint32_t *__restrict__ parent, uint32_t a, uint32_t b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }   // a < b: hang b's root under a
        const uint32_t old = atomicMin(&parent[b], a);
        if (old == b) return;                                // b was still a root: linked
        b = old;                                             // somebody re-parented b meanwhile: retry from there
    }
}

// 64 map bits from bit position q of the frame's map on (q may be negative: the bits in front of the map are 0; the map's rows are padded
// to whole tiles with zeros and `nwords` words long)
__device__ __forceinline__ uint64_t map_bits(const uint64_t *__restrict__ bm, uint32_t nwords, int64_t q)
{
    if (q <= -64) return 0;
    if (q < 0) return bm[0] << (uint32_t)(-q);
    const uint32_t idx = (uint32_t)(q >> 6), sh = (uint32_t)(q & 63);
    if (idx >= nwords) return 0;
    const uint64_t lo = bm[idx];
    if (!sh) return lo;
    const uint64_t hi = idx + 1 < nwords ? bm[idx + 1] : 0ull;
    return (lo >> sh) | (hi << (64u - sh));
}

// grid (ceil(ntiles / WAVES), B), a wavefront per tile
__global__ __launch_bounds__(WG) void k_l2_prep(Scratch sc, L2Work w)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t t = blockIdx.x * WAVES + wv, f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint64_t *bm = reinterpret_cast<const uint64_t *>(sc.bitmap + (uint64_t)f * sc.nb_stride);
    const uint64_t bits = bm[(uint64_t)t * 64 + lane];
    const uint32_t cnt = (uint32_t)__builtin_popcountll(bits);
    const uint32_t inc = wave_incl_scan(cnt);
    w.word_base[((uint64_t)f * sc.ntiles + t) * 64 + lane] = (uint16_t)(inc - cnt);   // (< 4096: the last word's base is at most 4032)
    const uint32_t total = wave_last(inc);
    const uint64_t fbase = (uint64_t)f * w.ids_per_frame;
    const uint32_t id0 = t * (uint32_t)TILE_PX;
    for (uint32_t r = lane; r < total; r += 64) {
        w.parent[fbase + id0 + r] = id0 + r;
        w.stat[fbase + id0 + r] = 0;
    }
}

// id of the set pixel at linear position p of the frame (the caller has seen its bit)
__device__ __forceinline__ uint32_t l2_id(const uint64_t *__restrict__ bm, const uint16_t *__restrict__ wbase, uint32_t p)
{
    const uint32_t wi = p >> 6;
    return (p >> 12) * (uint32_t)TILE_PX + wbase[wi] + (uint32_t)__builtin_popcountll(bm[wi] & ((1ull << (p & 63u)) - 1ull));
}

// grid (ceil(ntiles / WAVES), B), a wavefront per tile, a lane per word
__global__ __launch_bounds__(WG) void k_l2_link(Scratch sc, L2Work w, uint32_t nx)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t t = blockIdx.x * WAVES + wv, f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint32_t nwords = sc.ntiles * 64u;
    const uint64_t *bm = reinterpret_cast<const uint64_t *>(sc.bitmap + (uint64_t)f * sc.nb_stride);
    const uint16_t *wbase = w.word_base + (uint64_t)f * nwords;
    const uint32_t wi = t * 64u + (uint32_t)lane;
    const uint64_t W = bm[wi];
    if (!W) return;                       // (nothing below crosses lanes)
    const uint32_t p0 = wi * 64u;         // first pixel of the word (N < 2^32)
    // pixels of this word in the first / last column of their row
    uint64_t col0 = 0, colL = 0;
    for (uint64_t r = (((uint64_t)p0 + nx - 1) / nx) * nx; r < (uint64_t)p0 + 64; r += nx) col0 |= 1ull << (r - p0);        // row starts inside the word
    for (uint64_t r = (((uint64_t)p0 + nx) / nx) * nx; r < (uint64_t)p0 + 65; r += nx) colL |= 1ull << (r - 1 - p0);        // pixel r - 1 ends a row
    // neighbour maps, bit i = "pixel p0 + i has that neighbour set"
    const uint64_t west = map_bits(bm, nwords, (int64_t)p0 - 1) & ~col0;
    const uint64_t A = map_bits(bm, nwords, (int64_t)p0 - nx - 1), Bn = map_bits(bm, nwords, (int64_t)p0 - nx + 63);
    const uint64_t nw = A & ~col0, north = (A >> 1) | (Bn << 63), ne = ((A >> 2) | (Bn << 62)) & ~colL;
    uint32_t *parent = w.parent + (uint64_t)f * w.ids_per_frame;
    const uint32_t id0 = t * (uint32_t)TILE_PX + wbase[wi];
    for (uint64_t todo = W & (west | nw | north | ne); todo; todo &= todo - 1) {
        const uint32_t i = (uint32_t)__builtin_ctzll(todo);
        const uint64_t bit = 1ull << i;
        const uint32_t p = p0 + i, self = id0 + (uint32_t)__builtin_popcountll(W & (bit - 1ull));
        if (north & bit) uf_union(parent, self, l2_id(bm, wbase, p - nx));            // (W, NW, NE hang on N through their own links)
        else {
            if (west & bit) uf_union(parent, self, l2_id(bm, wbase, p - 1));          // (NW is W's northern neighbour)
            else if (nw & bit) uf_union(parent, self, l2_id(bm, wbase, p - nx - 1));
            if (ne & bit) uf_union(parent, self, l2_id(bm, wbase, p - nx + 1));
        }
    }
}

// use_sum: 0 = maximum, 1 = sum.  The sum wraps the way the reference's arithmetic would: its statistic is cast to the source dtype
// (recode_writer.py:446 hands `self._src_dtype` to get_summary_stats_nb) and stored in src_bit_depth bits (_bit_pack drops the bits above,
// recode_writer.py:637-652) - i.e. the sum modulo 2^d.  The accumulator is 32 bits wide (2^d divides 2^32: wrapping it changes nothing),
// k_l2_emit keeps its low 16 bits, the d-bit pack the low d.
__global__ __launch_bounds__(WG) void k_l2_stats(Scratch sc, L2Work w, uint32_t use_sum)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t t = blockIdx.x * WAVES + wv, f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint64_t frow = (uint64_t)f * sc.ntiles;
    const uint32_t total = sc.tile_cnt[frow + t];
    uint32_t *parent = w.parent + (uint64_t)f * w.ids_per_frame, *stat = w.stat + (uint64_t)f * w.ids_per_frame;
    const uint16_t *vals = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(sc.pix_slots) + (frow + t) * sc.pix_slot_bytes);
    const uint32_t id0 = t * (uint32_t)TILE_PX;
    for (uint32_t r = lane; r < total; r += 64) {
        const uint32_t id = id0 + r;
        if (parent[id] == id) continue;                      // a root (every pixel without neighbours): its own value joins at the emit
        const uint32_t root = uf_find(parent, id);
        parent[id] = root;
        if (use_sum) atomicAdd(&stat[root], (uint32_t)vals[r]);
        else atomicMax(&stat[root], (uint32_t)vals[r]);
    }
}

// The tile's roots -> its new value list (see the head of the file).  The workgroup's LDS holds a tile's worst case per wavefront.
__global__ __launch_bounds__(WG) void k_l2_emit(Scratch sc, L2Work w, uint32_t use_sum, uint32_t depth)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_list[WAVES][TILE_PX];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t t = blockIdx.x * WAVES + wv, f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint64_t frow = (uint64_t)f * sc.ntiles;
    const uint32_t total = sc.tile_cnt[frow + t];
    if (total == 0) return;               // (the tile stays empty)
    const uint32_t *parent = w.parent + (uint64_t)f * w.ids_per_frame, *stat = w.stat + (uint64_t)f * w.ids_per_frame;
    uint16_t *slot = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(sc.pix_slots) + (frow + t) * sc.pix_slot_bytes);
    uint16_t *list = s_list[wv];
    const uint32_t id0 = t * (uint32_t)TILE_PX;
    uint32_t nroot = 0;
    for (uint32_t r0 = 0; r0 < total; r0 += 64) {
        const uint32_t r = r0 + lane;
        uint32_t isroot = 0, v = 0;
        if (r < total) {
            const uint32_t id = id0 + r;
            isroot = parent[id] == id ? 1u : 0u;
            if (isroot) {
                const uint32_t own = slot[r], acc = stat[id];
                v = use_sum ? acc + own : max(acc, own);
            }
        }
        const uint32_t inc = wave_incl_scan(isroot);
        if (isroot) list[nroot + inc - 1] = (uint16_t)v;
        nroot += wave_last(inc);
    }
    __builtin_amdgcn_wave_barrier();
    if (depth < 16 && nroot) pack_stage(list, nroot, depth);
    __builtin_amdgcn_wave_barrier();
    // whole lines, like the reduce kernel's residual lines (k_gather reads whole 16-byte pieces of the stream)
    const uint32_t ndw = (((nroot * (depth < 16 ? depth : 16u) + 31) >> 5) + 31u) & ~31u;
    for (uint32_t i = lane; i < ndw; i += 64) reinterpret_cast<uint32_t *>(slot)[i] = reinterpret_cast<const uint32_t *>(list)[i];
    if (lane == 0) sc.tile_cnt[frow + t] = nroot;
}

void launch_l2(const Scratch &sc, const L2Work &w, uint32_t B, uint32_t nx, uint32_t use_sum, uint32_t depth, hipStream_t s)
{
    const dim3 grid((sc.ntiles + WAVES - 1) / WAVES, B);
    hipLaunchKernelGGL(k_l2_prep, grid, dim3(WG), 0, s, sc, w);
    hipLaunchKernelGGL(k_l2_link, grid, dim3(WG), 0, s, sc, w, nx);
    hipLaunchKernelGGL(k_l2_stats, grid, dim3(WG), 0, s, sc, w, use_sum);
    hipLaunchKernelGGL(k_l2_emit, grid, dim3(WG), 0, s, sc, w, use_sum, depth);
}

// ---- validation frames (reference recode_writer.py:402-415): the dose-rate count on the streaming path ---------------------
// Every validation_frame_gap-th frame the reference counts the 8-connected components of the binary map inside the central
// ROI (at most 128 x 128 pixels) with scipy.ndimage.label.  Here one workgroup per selected frame of the batch binarises the
// ROI straight from the frame (frame > thr, the pixels the reduce kernel has just read), labels it in LDS by minimum
// propagation over the 8 neighbours (plus one pointer jump per sweep) until nothing changes (a component's label converges to its smallest pixel index + 1:
// monotone, so unsynchronised in-place updates reach the same fixed point), and counts the pixels that kept their own
// label.  counts[i] = 0xFFFFFFFF for frames that are not validation frames.
constexpr int ROI_MAX = 128, ROI_T = 256;
__global__ __launch_bounds__(ROI_T) void k_roi_components(const void *__restrict__ frames, const void *__restrict__ thr_any, uint64_t N,
                                                           uint32_t nx, uint32_t first_frame_id, uint32_t gap, uint32_t x0, uint32_t y0,
                                                           uint32_t w, uint32_t h, uint32_t *__restrict__ counts, uint32_t src_bytes)
{
    __shared__ uint16_t lab[(ROI_MAX + 2) * (ROI_MAX + 2)];   // one cell of zero border all round
    __shared__ uint32_t s_cnt;
    const uint32_t i = blockIdx.x;
    if ((first_frame_id + i) % gap != 0) { if (threadIdx.x == 0) counts[i] = 0xFFFFFFFFu; return; }
    const uint16_t *fr = static_cast<const uint16_t *>(frames) + (uint64_t)i * N;      // (uint8 sources: fr8; uint32 sources: fr32 and a uint32 threshold)
    const uint8_t *fr8 = static_cast<const uint8_t *>(frames) + (uint64_t)i * N;
    const uint32_t *fr32 = static_cast<const uint32_t *>(frames) + (uint64_t)i * N;
    const uint16_t *thr = static_cast<const uint16_t *>(thr_any);
    const uint32_t *thr32 = static_cast<const uint32_t *>(thr_any);
    const uint32_t W = w + 2;
    for (uint32_t k = threadIdx.x; k < (h + 2) * W; k += ROI_T) lab[k] = 0;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < w * h; k += ROI_T) {
        const uint32_t yy = k / w, xx = k - yy * w;
        const uint64_t p = (uint64_t)(y0 + yy) * nx + (x0 + xx);
        const bool set = src_bytes == 4 ? fr32[p] > thr32[p] : (src_bytes == 1 ? (uint32_t)fr8[p] : (uint32_t)fr[p]) > thr[p];
        if (set) lab[(yy + 1) * W + xx + 1] = (uint16_t)(k + 1);
    }
    __syncthreads();
    for (uint32_t it = 0; it < (uint32_t)ROI_MAX * ROI_MAX; ++it) {   // (a component's diameter bounds the sweeps)
        int changed = 0;
        for (uint32_t k = threadIdx.x; k < w * h; k += ROI_T) {
            const uint32_t yy = k / w, xx = k - yy * w;
            const uint32_t c = (yy + 1) * W + xx + 1;
            const uint32_t own = lab[c];
            if (!own) continue;
            uint32_t m = own;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const uint32_t v = lab[(int)c + dy * (int)W + dx];
                    if (v && v < m) m = v;
                }
            {   // pointer jumping: a label names a pixel; that pixel's own label is at least as good (long thin components
                // then converge in about log(length) sweeps instead of length)
                const uint32_t q = m - 1, qy = q / w, qx = q - qy * w;
                const uint32_t v = lab[(qy + 1) * W + qx + 1];
                if (v && v < m) m = v;
            }
            if (m < own) { lab[c] = (uint16_t)m; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    uint32_t roots = 0;
    for (uint32_t k = threadIdx.x; k < w * h; k += ROI_T) {
        const uint32_t yy = k / w, xx = k - yy * w;
        roots += lab[(yy + 1) * W + xx + 1] == k + 1 ? 1u : 0u;
    }
    atomicAdd(&s_cnt, roots);
    __syncthreads();
    if (threadIdx.x == 0) counts[i] = s_cnt;
}
void launch_roi_components(const void *frames, const void *thr, uint64_t N, uint32_t nx, uint32_t n, uint32_t first_frame_id, uint32_t gap,
                           uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t *counts, hipStream_t s, uint32_t src_bytes)
{
    hipLaunchKernelGGL(k_roi_components, dim3(n), dim3(ROI_T), 0, s, frames, thr, N, nx, first_frame_id, gap, x0, y0, w, h, counts, src_bytes);
}

}  // namespace rc
