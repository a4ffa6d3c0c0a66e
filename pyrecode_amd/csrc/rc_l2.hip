// rc_l2.hip - reduction level 2: one summary statistic per connected component of the binary map.
//
// Reference intent (the reference's own L2 path cannot run): pyrecode/recode_writer.py:443-446
//   labeled, n = scipy.ndimage.label(binary_frame, structure=ones((3,3)))      8-connectivity, labels in raster order of each
//   stats = get_summary_stats_nb(labeled, frame, 0, dtype, 'max' | 'sum')      component's first pixel; statistic of the RAW
//   (pyrecode/utils/converters.py:262-297)                                     frame values, cast to the source dtype
// The record then carries the full binary map and the statistics where L1 carries the residuals (recode_writer.py:461-525).
//
// Device formulation (round 5).  The reduce kernel leaves, per tile of 4096 pixels, the raw binary map (64 words of 64 pixels), the raw
// values of its set pixels in raster order (the tile's value slot) - exactly what it left in rounds 2-4.  A
// set pixel's ID inside its frame is tile * 4096 + its rank among the tile's set pixels: ids grow in raster order, need no per-frame
// prefix (no scan in front of this stage), and index a sparse array of NODES {parent, accumulator} that rest at zero between batches -
// no pass initialises them, and only pixels with neighbours ever have theirs written.  Four passes; a work ITEM is 64 consecutive tiles
// of one frame and belongs to a one-wave workgroup (few fat waves instead of a quarter of a million one-tile waves: next to the following
// batch's reduce kernel a CU has room for a handful of small waves, and a pass of one-tile waves cost ~190 us however little they did):
//   k_l2_dir     the directory: for every 64-pixel word the set pixels of its tile in front of it (a lane per word, a wave scan per tile) -
//                what turns a neighbour's position into its id
//   k_l2_link    the item's NON-EMPTY words are listed first (ballot + rank: half the words at 1 %, one in sixteen at 0.1 %), then a
//                lane per listed word: which of its pixels have a set neighbour among W, NW, N, NE comes from WORD arithmetic on the
//                word, its left neighbour and the words one row up (funnel-shifted: the row length need not be a multiple of 64) - at
//                1 % Bernoulli 96 % of the set pixels have none and are done.  The others are listed in LDS and then taken a LANE
//                PER PIXEL: union with N, or with W / NW and NE (the rest of the four hang on those through their
//                own links); neighbour id = directory entry + a popcount; union-find with the smaller id as root (compare-and-swap links, path halving)
//   k_l2_stats   a lane per set pixel of the item (pixels numbered through, tile by binary search): a pixel that is not its own
//                root finds it and adds its raw value (atomicMax / atomicAdd on the root's accumulator)
//   k_l2_emit    the same walk: roots - in id order, i.e. in scipy's label order - leave their statistic (own value joined with what
//                the others added) in an ordered LDS list; then every tile's share of the list is bit-packed into the tile-local d-bit
//                stream exactly as the reduce kernel leaves level-1 residuals and written back to the tile's slot with the tile's
//                new count.  From there on the batch IS a level-1 batch: scans, record layout, k_gather and every codec run unchanged.
// Rounds 2-4 compacted every set pixel into batch-global arrays (position, value, parent, accumulator: 14 bytes each), looked every
// pixel's four neighbours up in the map one by one, and emitted per frame with ONE workgroup: 870 us per 64 frames of 4096^2 at 1 %
// (profiles/r05_exp4_level2_traces_dense_sweep.log), 1.5 ms on clustered events.
#include "rc_launch.h"

namespace rc {

__device__ __forceinline__ uint32_t *node_parent(u32x2 *node, uint32_t id) { return reinterpret_cast<uint32_t *>(node + id); }
__device__ __forceinline__ uint32_t *node_stat(u32x2 *node, uint32_t id) { return reinterpret_cast<uint32_t *>(node + id) + 1; }

// Nodes rest at {0, 0} between batches (zeroed once at allocation; k_l2_emit puts back what a batch changed): parent = 0 says "a root",
// anything else is the parent's id + 1.  No pass initialises them - and the nine set pixels in ten that have no neighbour never have
// theirs written at all.
// PLAIN: ordinary loads and stores (served by the XCD's own caches) instead of device-scope ones.  Right wherever no link is made meanwhile
// (k_l2_stats: the forest is final, every pointer a find can see - however stale - is an ancestor, and ancestors are forever).  While links
// ARE made (k_l2_link) the device-scope form stays: plain finds would be right there too - the compare-and-swap at the root validates
// them - but gained nothing (profiles/r05_exp28_level2_stats_streaks_plain_finds.log).
template <bool PLAIN>
__device__ __forceinline__ uint32_t uf_find_t(u32x2 *__restrict__ node, uint32_t x)
{
    uint32_t p = PLAIN ? *node_parent(node, x) : __hip_atomic_load(node_parent(node, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != 0) {
        const uint32_t gp = PLAIN ? *node_parent(node, p - 1) : __hip_atomic_load(node_parent(node, p - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != 0) {  // path halving
            if (PLAIN) *node_parent(node, x) = gp;
            else __hip_atomic_store(node_parent(node, x), gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        x = p - 1;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ uint32_t uf_find(u32x2 *__restrict__ node, uint32_t x) { return uf_find_t<false>(node, x); }
__device__ __attribute__((noinline)) void uf_union(u32x2 *__restrict__ node, uint32_t a, uint32_t b)
{
    for (;;) {
        a = uf_find(node, a);
        b = uf_find(node, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }   // a < b: hang b's root under a (the smaller id - the earlier pixel - stays root)
        const uint32_t old = atomicCAS(node_parent(node, b), 0u, a + 1u);
        if (old == 0) return;                                // b was still a root: linked
        b = old - 1;                                         // somebody hung b elsewhere meanwhile: go on from there
    }
}

__device__ __forceinline__ uint32_t l2_lane_get(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * src_lane), (int)v); }

// Rank bases: base[w] = set pixels of word w's tile in front of the word (w: word index inside the frame) - the directory a neighbour's id is
// looked up in.  k_l2_dir writes it for the whole batch before k_l2_link runs.  (Round 5 kept the bases of an item's 64 + 18 tiles in the
// item's LDS - 10.5 KB; with the word and pixel lists a k_l2_link workgroup held 16.6 KB and k_l2_emit 12.8 KB.  The reduce kernel of the next
// batch fills the CUs' LDS up to 32 KB and their registers up to 104 per SIMD: a second-stage workgroup that needs more of either than is
// free waits for a reduce workgroup to retire and then runs IN ITS PLACE.  Padding these workgroups with another 16 KB cost the step
// 19 %, profiles/r05_exp20_level2_lds_room.log.  And the compiler raises a kernel's register claim to 129 when its LDS alone limits it to
// three waves per SIMD.  Hence: every kernel of this file within 2.6 KB of LDS and 64 registers.)
struct L2View { const uint64_t *bm; const uint16_t *base; };
// id of the set pixel at linear position p of the frame (the caller has seen its bit)
__device__ __forceinline__ uint32_t l2_id(const L2View &v, uint32_t p)
{
    const uint32_t wi = p >> 6;
    return (p >> 12) * (uint32_t)TILE_PX + v.base[wi] + (uint32_t)__builtin_popcountll(v.bm[wi] & ((1ull << (p & 63u)) - 1ull));
}
// pixel p (id self) joins the earlier ones among its upper neighbours: N, or NW and NE (its W neighbour, if it has one, is its run's business:
// k_l2_dir; the rest hang on these through their own links)
__device__ __forceinline__ void l2_link_pixel(u32x2 *node, const L2View &v, uint32_t nx, uint32_t p, uint32_t self, bool n, bool nw, bool ne)
{
    if (n) uf_union(node, self, l2_id(v, p - nx));
    else {
        if (nw) uf_union(node, self, l2_id(v, p - nx - 1));
        if (ne) uf_union(node, self, l2_id(v, p - nx + 1));
    }
}

// grid: one-wave workgroups over items of 64 tiles of one frame; a lane per 64-pixel word, one wave scan per tile, eight tiles in flight.
// The same pass hangs every pixel that has a W neighbour on the first pixel of its horizontal RUN (a plain store: no other kernel runs yet,
// nothing races) - or, where the run comes in from the word in front, on that word's last pixel, which hangs on ITS run's first.  What is
// left for k_l2_link are the links between rows: of a run's first pixel, and of a later pixel only where it has a NE neighbour and no N (its
// W neighbour has every other upper neighbour of its own among its neighbours).  A 4 x 4 blob: 3 unions instead of 15.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void k_l2_dir(Scratch sc, uint32_t nx, uint32_t gpf, uint32_t nitems)
{
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t step = (uint32_t)TILE_PX % nx;     // a tile further on is this much further along its row (mod nx)
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint32_t f = item / gpf, t0 = 64u * (item - f * gpf);
        const uint64_t frow = (uint64_t)f * sc.ntiles;
        const uint64_t *bm = reinterpret_cast<const uint64_t *>(sc.bitmap + (uint64_t)f * sc.nb_stride);
        uint16_t *base = sc.l2_base + frow * 64u;
        u32x2 *node = sc.l2_node + (uint64_t)f * sc.l2_ids_per_frame;
        const uint32_t ntl = min(64u, sc.ntiles - t0);
        uint32_t x0 = (uint32_t)(((uint64_t)t0 * TILE_PX + 64u * lane) % nx);   // column of the word's first pixel
        uint32_t last_hi = t0 > 0 ? (uint32_t)(bm[t0 * 64u - 1u] >> 63) : 0u;  // (uniform) the last pixel of the tile in front: one load per item
        for (uint32_t c0 = 0; c0 < ntl; c0 += 8) {
            uint64_t W[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) W[k] = c0 + k < ntl ? bm[(t0 + c0 + k) * 64u + lane] : 0ull;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (c0 + k >= ntl) break;                       // (uniform)
                const uint32_t tile = t0 + c0 + k;
                const uint64_t Wd = W[k];
                const uint32_t cnt = (uint32_t)__builtin_popcountll(Wd);
                const uint32_t b = wave_incl_scan(cnt) - cnt;
                base[tile * 64u + lane] = (uint16_t)b;
                const uint32_t hi = (uint32_t)(Wd >> 63);
                uint32_t pm = wave_prev(hi);                     // the last pixel of the word in front
                if (lane == 0) pm = last_hi;
                last_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
                uint64_t col0 = 0;                               // pixels of this word in the first column of their row: no W neighbour
                for (uint32_t i = x0 ? nx - x0 : 0u; i < 64; i += nx) col0 |= 1ull << i;
                const uint64_t Lk = ((Wd << 1) | pm) & ~col0;    // bit i: pixel i - 1 is set and in the same row
                uint64_t nh = Wd & Lk;
                if (nh) {
                    const uint64_t H = Wd & ~Lk;                 // the runs' first pixels
                    const uint32_t id0 = tile * (uint32_t)TILE_PX + b;
                    // the set pixel in front of this word's first: one id down, or the previous tile's last
                    const uint32_t prev_id = (nh & 1ull) && b == 0 ? (tile - 1u) * (uint32_t)TILE_PX + sc.tile_cnt[frow + tile - 1u] - 1u : id0 - 1u;
                    for (; nh; nh &= nh - 1) {
                        const uint32_t i = (uint32_t)__builtin_ctzll(nh);
                        const uint64_t below = (1ull << i) - 1ull, hb = H & below;
                        const uint32_t par = hb ? id0 + (uint32_t)__builtin_popcountll(Wd & ((1ull << (63 - __builtin_clzll(hb))) - 1ull)) : prev_id;
                        *node_parent(node, id0 + (uint32_t)__builtin_popcountll(Wd & below)) = par + 1u;
                    }
                }
                x0 += step;
                if (x0 >= nx) x0 -= nx;
            }
        }
    }
}

constexpr uint32_t L2_WORDS = 512;   // the words of a chunk of 8 tiles: its non-empty ones are listed in LDS (uint16 each)
constexpr uint32_t L2_DESC = 256;    // linked pixels listed per round (LDS, one dword each)

// grid: one-wave workgroups over the items (64 tiles of one frame each)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void k_l2_link(Scratch sc, uint32_t nx, uint32_t gpf, uint32_t nitems)
{
    __shared__ uint16_t s_words[L2_WORDS];
    __shared__ uint32_t s_desc[L2_DESC];
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t nwords = sc.ntiles * 64u;
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint32_t f = item / gpf, t0 = 64u * (item - f * gpf);
        const uint64_t *bm = reinterpret_cast<const uint64_t *>(sc.bitmap + (uint64_t)f * sc.nb_stride);
        u32x2 *node = sc.l2_node + (uint64_t)f * sc.l2_ids_per_frame;
        const uint32_t ntl = min(64u, sc.ntiles - t0);
        const L2View view{bm, sc.l2_base + (uint64_t)f * sc.ntiles * 64u};
        uint32_t listed = 0;
        auto drain = [&]() {   // desc = word index inside the item (12 bits) << 10 | bit (6) << 4 | N (8) NW (2) NE (1)
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = lane; j < listed; j += 64) {
                const uint32_t d = s_desc[j];
                const uint32_t p = ((t0 * 64u + (d >> 10)) << 6) + ((d >> 4) & 63u);
                l2_link_pixel(node, view, nx, p, l2_id(view, p), (d & 8u) != 0, (d & 2u) != 0, (d & 1u) != 0);
            }
            __builtin_amdgcn_wave_barrier();
            listed = 0;
        };
        for (uint32_t c0 = 0; c0 < ntl; c0 += 8) {
            // ---- A: the chunk's NON-EMPTY words (half of them at 1 % of the pixels set, one in sixteen at 0.1 %) listed by ballot + rank
            uint32_t nw = 0;
            uint64_t W[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) W[k] = c0 + k < ntl ? bm[(t0 + c0 + k) * 64u + lane] : 0ull;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint64_t m = __builtin_amdgcn_ballot_w64(W[k] != 0);
                if (W[k]) s_words[nw + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)((c0 + k) * 64u + lane);
                nw += (uint32_t)__builtin_popcountll(m);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- B: a lane per non-empty word of the chunk - which of its pixels have a set neighbour among W, NW, N, NE; C: those pixels, a lane each
            for (uint32_t j0 = 0; j0 < nw; j0 += 64) {
                const bool have = j0 + lane < nw;
                const uint32_t wrel = have ? s_words[j0 + lane] : 0u;
                const uint32_t wi = t0 * 64u + wrel;
                const uint64_t p0 = (uint64_t)wi * 64;
                const int64_t q = (int64_t)p0 - (int64_t)nx - 1;          // the map bit of pixel 0's NW neighbour
                const int64_t idx = q >> 6;                                // (floor: q may be negative)
                const uint32_t sh = (uint32_t)(q & 63);
                uint64_t Wd = 0, Wl = 0, n0 = 0, n1 = 0, n2 = 0;
                if (have) {
                    Wd = bm[wi];
                    Wl = wi > 0 ? bm[wi - 1] : 0ull;
                    n0 = idx >= 0 && idx < (int64_t)nwords ? bm[idx] : 0ull;
                    n1 = idx + 1 >= 0 && idx + 1 < (int64_t)nwords ? bm[idx + 1] : 0ull;
                    n2 = sh >= 62 && idx + 2 >= 0 && idx + 2 < (int64_t)nwords ? bm[idx + 2] : 0ull;   // (only bits 0 and 1 of the continuation are used)
                }
                // A: bit i = pixel p0 + i - nx - 1 (NW); Bn: its continuation, for N and NE
                const uint64_t A = sh ? (n0 >> sh) | (n1 << (64u - sh)) : n0;
                const uint64_t Bn = sh ? (n1 >> sh) | (n2 << (64u - sh)) : n1;
                uint64_t col0 = 0, colL = 0;   // pixels of this word in the first / last column of their row
                if (have) {
                    for (uint64_t r = ((p0 + nx - 1) / nx) * nx; r < p0 + 64; r += nx) col0 |= 1ull << (r - p0);
                    for (uint64_t r = ((p0 + nx) / nx) * nx; r < p0 + 65; r += nx) colL |= 1ull << (r - 1 - p0);
                }
                const uint64_t fW = ((Wd << 1) | (Wl >> 63)) & ~col0, fNW = A & ~col0, fN = (A >> 1) | (Bn << 63), fNE = ((A >> 2) | (Bn << 62)) & ~colL;
                // a pixel with a W neighbour hangs on its run already (k_l2_dir) and needs a link of its own only to a NE neighbour its W neighbour
                // cannot see (no N between them)
                uint64_t todo = Wd & ((~fW & (fNW | fN | fNE)) | (fW & fNE & ~fN));
                const uint32_t n = (uint32_t)__builtin_popcountll(todo);
                const uint32_t inc = wave_incl_scan(n), total = wave_last(inc);
                if (total == 0) continue;
                const bool fits = total <= L2_DESC;            // (dense maps: 64 words can hold more linked pixels than the list - then every lane links its own)
                if (fits && listed + total > L2_DESC) drain();
                uint32_t o = listed + inc - n;
                const uint32_t id0 = (wi >> 6) * (uint32_t)TILE_PX + ((fits || !todo) ? 0u : (uint32_t)view.base[wi]);
                for (; todo; todo &= todo - 1) {
                    const uint32_t i = (uint32_t)__builtin_ctzll(todo);
                    const uint64_t bit = 1ull << i;
                    const bool nwl = (fNW & ~fW & bit) != 0;
                    if (fits) s_desc[o++] = (wrel << 10) | (i << 4) | ((fN & bit) ? 8u : 0u) | (nwl ? 2u : 0u) | ((fNE & bit) ? 1u : 0u);
                    else l2_link_pixel(node, view, nx, wi * 64u + i, id0 + (uint32_t)__builtin_popcountll(Wd & (bit - 1ull)), (fN & bit) != 0, nwl, (fNE & bit) != 0);
                }
                if (fits) listed += total;
            }
            __builtin_amdgcn_wave_barrier();
        }
        drain();
        __builtin_amdgcn_wave_barrier();
    }
}

// The item's set pixels numbered through: cum = exclusive wave scan of the tiles' counts (a lane per tile); pixel P belongs to the last tile
// whose first pixel is <= P.
__device__ __forceinline__ uint32_t l2_tile_of(uint32_t cum, uint32_t P)
{
    uint32_t k = 0;
#pragma unroll
    for (uint32_t step = 32; step; step >>= 1) {
        const uint32_t c = l2_lane_get(cum, k + step);
        if (c <= P) k += step;
    }
    return k;
}

// use_sum: 0 = maximum, 1 = sum.  The sum wraps the way the reference's arithmetic would: its statistic is cast to the source dtype
// (recode_writer.py:446 hands `self._src_dtype` to get_summary_stats_nb) and stored in src_bit_depth bits (_bit_pack drops the bits above,
// recode_writer.py:637-652) - i.e. the sum modulo 2^d.  The accumulator is 32 bits wide (2^d divides 2^32: wrapping it changes nothing),
// k_l2_emit keeps its low 16 bits, the d-bit pack the low d.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void k_l2_stats(Scratch sc, uint32_t use_sum, uint32_t gpf, uint32_t nitems)
{
    const uint32_t lane = (uint32_t)lane_id();
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint32_t f = item / gpf, t0 = 64u * (item - f * gpf), t = t0 + lane;
        const uint64_t frow = (uint64_t)f * sc.ntiles;
        const uint32_t cnt = t < sc.ntiles ? sc.tile_cnt[frow + t] : 0u;
        const uint32_t inc = wave_incl_scan(cnt), cum = inc - cnt, T = wave_last(inc);
        u32x2 *node = sc.l2_node + (uint64_t)f * sc.l2_ids_per_frame;
        const uint8_t *slots = reinterpret_cast<const uint8_t *>(sc.pix_slots) + (frow + t0) * sc.pix_slot_bytes;
        constexpr int U = 4;
        for (uint32_t i0 = 0; 64u * i0 < T; i0 += U) {
            uint32_t id[U], par[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t P = 64u * (i0 + u) + lane;
                const uint32_t k = l2_tile_of(cum, P), r = P - l2_lane_get(cum, k);
                id[u] = P < T ? (t0 + k) * (uint32_t)TILE_PX + r : 0xFFFFFFFFu;
                par[u] = P < T ? node[id[u]][0] : 0u;
            }
            // A pixel that is not its own root finds it (plain loads: the forest is final) - and consecutive lanes that found the SAME root (the
            // pixels of a horizontal run are consecutive ids) join their values first: one atomic per streak instead of one per pixel (the
            // accumulators are device-scope atomics, ~19 G/s for the whole device; a blob of 16 pixels sent 15 of them to one address)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool hung = par[u] != 0;
                const uint64_t hb = __builtin_amdgcn_ballot_w64(hung);
                if (hb == 0) continue;                                         // (uniform)
                uint32_t root = 0xFFFFFFFFu, v = 0;
                if (hung) {
                    root = uf_find_t<true>(node, id[u]);
                    v = reinterpret_cast<const uint16_t *>(slots + (uint64_t)((id[u] >> 12) - t0) * sc.pix_slot_bytes)[id[u] & (TILE_PX - 1)];
                }
                if (__builtin_popcountll(hb) <= 8) {                           // (uniform) a few scattered ones - 1 % Bernoulli: nothing to join
                    if (hung) {
                        if (use_sum) atomicAdd(node_stat(node, root), v);
                        else atomicMax(node_stat(node, root), v);
                    }
                    continue;
                }
                // segmented inclusive scan over the lanes, a segment = a streak of equal roots
                const uint32_t before = l2_lane_get(root, lane ? lane - 1u : 0u);
                uint32_t head = (lane == 0 || before != root) ? 1u : 0u;
#pragma unroll
                for (uint32_t st = 1; st < 64; st <<= 1) {
                    const uint32_t ov = l2_lane_get(v, lane >= st ? lane - st : lane), oh = l2_lane_get(head, lane >= st ? lane - st : lane);
                    if (lane >= st && !head) { v = use_sum ? v + ov : max(v, ov); head = oh; }
                }
                const uint32_t after = l2_lane_get(root, lane < 63 ? lane + 1u : lane);
                if (hung && (lane == 63 || after != root)) {              // the streak's last lane has its total
                    if (use_sum) atomicAdd(node_stat(node, root), v);
                    else atomicMax(node_stat(node, root), v);
                }
            }
        }
    }
}

constexpr uint32_t L2_ROUND = 1024;            // pixels walked per round: whole tiles while they fit, a big tile's pixels 1024 at a time
constexpr uint32_t L2_LIST = L2_ROUND + 32;    // roots' statistics listed per round (LDS, uint16 each) + what a big tile's round left over

// the tile-local stream of the d-bit fields of list[0 .. n) -> out; d = 16: the values.  pad: whole 128-byte lines, zero behind the stream
// (the end of a tile's stream); otherwise exactly the n * d / 32 dwords (n a multiple of 32: the stream goes on at a dword boundary)
__device__ __forceinline__ void l2_write_stream(const uint16_t *list, uint32_t n, uint32_t d, uint32_t *__restrict__ out, bool pad)
{
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t nbits = n * d, ndw = pad ? (((nbits + 31) >> 5) + 31u) & ~31u : nbits >> 5;
    const uint32_t dmask = d >= 16 ? 0xFFFFu : (1u << d) - 1u;
    for (uint32_t w = lane; w < ndw; w += 64) {
        uint32_t v = (32u * w) / d;
        const uint32_t o = 32u * w - v * d;
        uint64_t acc = 0;
        uint32_t filled = 0;
        if (v < n) { acc = (list[v] & dmask) >> o; filled = d - o; ++v; }
        while (filled < 32 && v < n) {
            acc |= (uint64_t)(list[v] & dmask) << filled;
            filled += d;
            ++v;
        }
        out[w] = (uint32_t)acc;
    }
}

// one pixel of the emit walk: its node (put back to rest if the batch touched it) and its raw value
__device__ __forceinline__ void l2_take(u32x2 *node, uint32_t id, const uint16_t *vals, uint32_t r, u32x2 &nd, uint32_t &own)
{
    nd = node[id];
    own = vals[r];
    if (nd[0] | nd[1]) node[id] = u32x2{0u, 0u};
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void k_l2_emit(Scratch sc, uint32_t use_sum, uint32_t depth, uint32_t gpf, uint32_t nitems)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_list[L2_LIST];
    __shared__ uint32_t s_n[64];                  // per tile of the round: how many roots,
    __shared__ uint16_t s_first[64];              // where they start in the list
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t d = depth < 16 ? depth : 16u;
    constexpr int U = 4;
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint32_t f = item / gpf, t0 = 64u * (item - f * gpf), t = t0 + lane;
        const uint64_t frow = (uint64_t)f * sc.ntiles;
        const uint32_t cnt = t < sc.ntiles ? sc.tile_cnt[frow + t] : 0u;
        const uint32_t inc = wave_incl_scan(cnt), cum = inc - cnt, T = wave_last(inc);
        if (T == 0) continue;
        u32x2 *node = sc.l2_node + (uint64_t)f * sc.l2_ids_per_frame;
        uint8_t *slots = reinterpret_cast<uint8_t *>(sc.pix_slots) + (frow + t0) * sc.pix_slot_bytes;
        // rounds of whole tiles whose pixels fit the list (25 tiles a round at 1 % of the pixels set, all 64 up to 0.4 %)
        for (uint32_t ka = 0; ka < 64;) {
            // kb: one behind the round's last tile = the first tile whose end lies more than L2_ROUND pixels behind tile ka's start
            const uint32_t start = l2_lane_get(cum, ka);
            const uint64_t over = __builtin_amdgcn_ballot_w64(lane >= ka && inc - start > L2_ROUND);
            const uint32_t kb = over ? (uint32_t)__builtin_ctzll(over) : 64u;
            if (kb == ka) {
                // ---- a tile with more pixels than a round walks: alone, L2_ROUND pixels at a time.  A round lists its roots behind the (< 32) the
                // round before left over and writes the largest multiple of 32 of them - whole dwords whatever d is, so the stream goes on at a dword
                // boundary; the last round writes the rest and the zeros up to the line's end.  In place: the stream never reaches the values
                // still to be read (no more roots than pixels, no more than 16 bits each).
                const uint32_t n_px = l2_lane_get(cnt, ka), id0 = (t0 + ka) * (uint32_t)TILE_PX;
                uint8_t *slot = slots + (uint64_t)ka * sc.pix_slot_bytes;
                const uint16_t *vals = reinterpret_cast<const uint16_t *>(slot);
                uint32_t rem = 0, written = 0;
                for (uint32_t r0 = 0; r0 < n_px; r0 += L2_ROUND) {
                    const uint32_t r1 = min(n_px, r0 + L2_ROUND);
                    uint32_t nlist = rem;
                    for (uint32_t q0 = r0; q0 < r1; q0 += 64u * U) {
                        u32x2 nd[U];
                        uint32_t own[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t r = q0 + 64u * u + lane;
                            nd[u] = u32x2{1u, 0u};
                            own[u] = 0;
                            if (r < r1) l2_take(node, id0 + r, vals, r, nd[u], own[u]);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const bool isroot = nd[u][0] == 0;          // (lanes behind r1 carry parent 1)
                            const uint32_t ri = wave_incl_scan(isroot ? 1u : 0u);
                            if (isroot) s_list[nlist + ri - 1] = (uint16_t)(use_sum ? nd[u][1] + own[u] : max(nd[u][1], own[u]));
                            nlist += wave_last(ri);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const bool last = r1 == n_px;
                    const uint32_t n_out = last ? nlist : nlist & ~31u;
                    l2_write_stream(s_list, n_out, d, reinterpret_cast<uint32_t *>(slot) + ((written * d) >> 5), last);
                    rem = nlist - n_out;
                    const uint16_t keep = lane < rem ? s_list[n_out + lane] : (uint16_t)0;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rem) s_list[lane] = keep;
                    __builtin_amdgcn_wave_barrier();
                    written += n_out;
                }
                if (lane == 0) sc.tile_cnt[frow + t0 + ka] = written;
                ka += 1;
                continue;
            }
            const uint32_t end = kb < 64 ? l2_lane_get(cum, kb) : T;
            s_n[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            uint32_t nlist = 0;
            for (uint32_t P0 = start; P0 < end; P0 += 64u * U) {
                uint32_t k[U], r[U];
                u32x2 nd[U];
                uint32_t own[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t P = P0 + 64u * u + lane;
                    k[u] = l2_tile_of(cum, P);
                    r[u] = P - l2_lane_get(cum, k[u]);
                    nd[u] = u32x2{1u, 0u};
                    own[u] = 0;
                    if (P < end) l2_take(node, (t0 + k[u]) * (uint32_t)TILE_PX + r[u], reinterpret_cast<const uint16_t *>(slots + (uint64_t)k[u] * sc.pix_slot_bytes), r[u], nd[u], own[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t P = P0 + 64u * u + lane;
                    const bool isroot = P < end && nd[u][0] == 0;
                    const uint32_t ri = wave_incl_scan(isroot ? 1u : 0u);
                    const uint32_t at = nlist + ri - 1;          // (roots only)
                    if (isroot) {
                        s_list[at] = (uint16_t)(use_sum ? nd[u][1] + own[u] : max(nd[u][1], own[u]));
                        atomicAdd(&s_n[k[u]], 1u);
                    }
                    if (P < end && r[u] == 0) s_first[k[u]] = (uint16_t)(at + (isroot ? 0u : 1u));   // the tile's first pixel: its roots start here (the next root's place)
                    nlist += wave_last(ri);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // every tile of the round: its share of the list -> its slot, as the tile-local packed stream; its new count.  A tile with at most eight roots whose fields fit 16 bytes (every tile at 0.1 % of the pixels
            // set) is finished by ITS OWN lane with one 16-byte store - all such tiles in one instruction; the others one after the other by
            // the whole wave
            const bool mine = lane >= ka && lane < kb && cnt != 0;
            const uint32_t my_n = mine ? s_n[lane] : 0u, my_first = mine ? s_first[lane] : 0u;
            const bool small = mine && my_n <= 8u && my_n * d <= 128u;
            if (small) {
                const uint32_t dmask = d >= 16 ? 0xFFFFu : (1u << d) - 1u;
                uint64_t lo = 0, hi = 0;
                for (uint32_t j = 0; j < my_n; ++j) {
                    const uint64_t v = s_list[my_first + j] & dmask;
                    const uint32_t b = j * d;
                    if (b < 64) { lo |= v << b; if (b + d > 64) hi |= v >> (64 - b); }
                    else hi |= v << (b - 64);
                }
                *reinterpret_cast<u32x4 *>(slots + (uint64_t)lane * sc.pix_slot_bytes) = u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                sc.tile_cnt[frow + t] = my_n;
            }
            __builtin_amdgcn_wave_barrier();
            for (uint64_t rest = __builtin_amdgcn_ballot_w64(mine && !small); rest; rest &= rest - 1) {
                const uint32_t kk = (uint32_t)__builtin_ctzll(rest);
                const uint32_t n = s_n[kk], first = s_first[kk];
                l2_write_stream(s_list + first, n, d, reinterpret_cast<uint32_t *>(slots + (uint64_t)kk * sc.pix_slot_bytes), true);
                if (lane == 0) sc.tile_cnt[frow + t0 + kk] = n;
            }
            __builtin_amdgcn_wave_barrier();
            ka = kb;
        }
    }
}

void launch_l2(const Scratch &sc, uint32_t B, uint32_t nx, uint32_t use_sum, uint32_t depth, hipStream_t s)
{
    const uint32_t gpf = (sc.ntiles + 63) / 64, nitems = gpf * B;
    static const char *wgs_env = RC_KNOB("RC_L2_WGS"), *lds_env = RC_KNOB("RC_L2_DYNLDS");   // (experiments: persistent grids, extra LDS per workgroup)
    const uint32_t grid = wgs_env ? min(nitems, (uint32_t)atoi(wgs_env)) : nitems, dyn = lds_env ? (uint32_t)atoi(lds_env) : 0u;
    hipLaunchKernelGGL(k_l2_dir, dim3(grid), dim3(64), 0, s, sc, nx, gpf, nitems);
    hipLaunchKernelGGL(k_l2_link, dim3(grid), dim3(64), dyn, s, sc, nx, gpf, nitems);
    hipLaunchKernelGGL(k_l2_stats, dim3(grid), dim3(64), 0, s, sc, use_sum, gpf, nitems);
    hipLaunchKernelGGL(k_l2_emit, dim3(grid), dim3(64), dyn, s, sc, use_sum, depth, gpf, nitems);
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
