// rc_reduce.hip - threshold/binarise/compact kernel, scans, record layout and record assembly (gfx950).
//
// Reference stages restated here (paths relative to the reference repo):
//   A1 thr = dark + eps                         pyrecode/recode_writer.py:126-137
//   A2 binary = frame > thr                     pyrecode/recode_writer.py:437
//   A3 pix = frame[binary] - thr[binary]        pyrecode/recode_writer.py:440
//   A4 LSB-first bitmap                         pyrecode/recode_writer.py:622-634
//   A5 LSB-first d-bit pack                     pyrecode/recode_writer.py:637-652
//   A7 record assembly                          pyrecode/recode_writer.py:485-494,518-525,546-550,559-574
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "rc_launch.h"
#include "rc_record.h"
#include "rc_pack.h"
#ifdef RC_PHASE_TIMING
__device__ unsigned long long g_phase[16];
#define RC_LZ4_PHASE 1   // (rc_lz4_block.h: sub-phases of the LZ4 encoder into g_phase[8..13])
#endif
#include "rc_lz4_block.h"
#include "rc_zstd_wave.h"
#include "rc_deflate_block.h"

namespace rc {

// Profiling builds only (tools/build_ablate.sh): -DRC_ABLATE=<bits> drops one kind of global store of the reduce kernel while
// keeping every computation alive (a value is "stored" only if it equals a magic number).
//   1: residual lines   2: encoded block lines   4: the 4-byte count / size stores
// -DRC_PHASE_TIMING (tools/build_def.sh): lane 0 of every wavefront adds the s_memtime cycles of each phase of reduce_one_frame to
// g_phase[] (rc_debug_phases reads and clears them).  s_memtime waits for the wave's outstanding LDS / scalar traffic, so the phases do
// not overlap as they do in the product build: the SHARES are what the numbers are good for.
#ifdef RC_PHASE_TIMING
// (sampled: one workgroup in 64 records, each wave's sums leave in one burst at the end of reduce_one_frame's last phase)
#define RC_PHASE_BEGIN unsigned long long ph_t_ = __builtin_amdgcn_s_memtime(); unsigned long long ph_a_[8] = {};
#define RC_PHASE(i) do { const unsigned long long ph_n_ = __builtin_amdgcn_s_memtime(); ph_a_[i] = ph_n_ - ph_t_; ph_t_ = ph_n_; \
        if ((i) == 6 && lane_id() == 0 && (blockIdx.x & 63u) == 0) { for (int q_ = 0; q_ < 7; ++q_) atomicAdd(&g_phase[q_], ph_a_[q_]); atomicAdd(&g_phase[7], 1ull); } } while (0)
#else
#define RC_PHASE_BEGIN
#define RC_PHASE(i) do { } while (0)
#endif
#ifndef RC_BZ
#define RC_BZ 4   // frames a wavefront keeps its tile (and the threshold registers) for
#endif
#ifndef RC_ABLATE
#define RC_ABLATE 0
#endif
#ifndef RC_FUSED12
#define RC_FUSED12 1   // d = 12: compaction and bit packing in one step (compact_pack12); 0 = compact, then pack_stage (A/B builds)
#endif
constexpr bool FUSED12 = RC_FUSED12 != 0;
__device__ uint32_t g_ablate_sink;
#define RC_ST(bit, lhs, v) do { if (RC_ABLATE & (bit)) { if ((uint32_t)(v) == 0x9E3779B9u) g_ablate_sink = 1; } else { lhs = (v); } } while (0)

__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b)
{
    uint32_t r;  // per 16-bit half: max(a - b, 0)  ==  (a > b) ? a - b : 0
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- A1 -------------------------------------------------------------------------------------------------
__global__ void k_threshold(const uint16_t *__restrict__ dark, uint32_t eps16, uint64_t N, uint16_t *__restrict__ thr)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < N; i += stride) thr[i] = (uint16_t)(dark[i] + eps16);
}

// uint8 sources: the sum wraps in the source dtype (numpy 2: uint8 + python int stays uint8, recode_writer.py:127); the device keeps
// thresholds as uint16 whatever the source
__global__ void k_threshold8(const uint8_t *__restrict__ dark, uint32_t eps8, uint64_t N, uint16_t *__restrict__ thr)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < N; i += stride) thr[i] = (uint16_t)(uint8_t)(dark[i] + eps8);
}

void launch_threshold(const void *dark, int64_t eps, uint64_t N, uint16_t *thr, hipStream_t s, uint32_t src_bytes)
{
    uint32_t blocks = (uint32_t)((N + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (src_bytes == 1)
        hipLaunchKernelGGL(k_threshold8, dim3(blocks), dim3(256), 0, s, static_cast<const uint8_t *>(dark), (uint32_t)((uint64_t)eps & 0xFF), N, thr);
    else
        hipLaunchKernelGGL(k_threshold, dim3(blocks), dim3(256), 0, s, static_cast<const uint16_t *>(dark), (uint32_t)((uint64_t)eps & 0xFFFF), N, thr);
}

// ---- A2+A3+A4: one pass over the frames ------------------------------------------------------------------
// Load 8 pixels (one bitmap byte) for this lane at pixel index px0; out-of-frame pixels read as `fill`.
// SB = bytes per source pixel: 2 (uint16 frames, the reference's use_c restriction and every BASELINE configuration) or 1 (uint8
// frames: source_bit_depth <= 8, reference misc.py:41-49).  A lane's 8 pixels of a group are one 16-byte / one 8-byte register set.
template <int SB> struct Src;
template <> struct Src<2> { typedef uint16_t T; typedef u32x4 X; };
template <> struct Src<1> { typedef uint8_t T; typedef u32x2 X; };
// 8 uint8 pixels -> the same 8 pixels as packed uint16 pairs (what the uint16 path loads)
__device__ __forceinline__ u32x4 widen8(const u32x2 &v)
{
    return u32x4{__builtin_amdgcn_perm(0u, v[0], 0x0C010C00u), __builtin_amdgcn_perm(0u, v[0], 0x0C030C02u),
                 __builtin_amdgcn_perm(0u, v[1], 0x0C010C00u), __builtin_amdgcn_perm(0u, v[1], 0x0C030C02u)};
}
__device__ __forceinline__ u32x4 widen8(const u32x4 &v) { return v; }

template <bool ALIGNED, bool STREAM, int SB = 2>
__device__ __forceinline__ typename Src<SB>::X load8(const typename Src<SB>::T *__restrict__ base, uint64_t px0, uint64_t N, uint32_t fill)
{
    typedef typename Src<SB>::X X;
    constexpr int PER = 4 / SB;   // pixels per dword
    // (the guarded form: a group of eight pixels that lies wholly inside the frame is one vector load as well; single loads with a
    // bounds check only for the group the frame ends in)
    if (ALIGNED || px0 + 8 <= N) {
        if (px0 < N) {
            // frames are read exactly once (nontemporal); the threshold tile is shared by other workgroups (cached).  The vector type is
            // declared with the pixel's alignment: a frame may start on any pixel boundary (amdhsa runs gfx9+ in unaligned access mode,
            // the compiler itself emits global_load_dwordx4 for such a load)
            typedef X XU __attribute__((aligned(SB)));
            const XU *p = reinterpret_cast<const XU *>(base + px0);
            return STREAM ? __builtin_nontemporal_load(p) : *p;
        }
    }
    X v;
#pragma unroll
    for (int j = 0; j < 8 / PER; ++j) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const uint64_t k = px0 + (uint64_t)(PER * j + e);
            const uint32_t val = (!ALIGNED && k < N) ? (uint32_t)base[k] : fill;
            w |= val << (8 * SB * e);
        }
        v[j] = w;
    }
    return v;
}

// Results of one (tile, frame) sitting in wave-private LDS / registers until flush_pending writes them out.  The stores are
// issued at the END of the frame's processing, i.e. behind the next frame's loads (which go out right after the subtract).
// gfx9 counts loads and stores on ONE in-order counter (vmcnt), so the wait in front of the next frame's first use of its data
// needs to cover the loads only: `s_waitcnt vmcnt(k)` with k = the number of store instructions issued behind them.  The
// compiler cannot know k (the stores sit in loops) and waits with vmcnt(0), i.e. for the stores as well - under a saturated
// read stream a store's acknowledgement takes as long as a load, and that whole latency was exposed once per frame
// (0.445 ms; 0.324 ms with every store dropped, everything else kept: tools/prof_ablate.sh, profiles/r02_reduce_stores.md).
// So the steady-state loads are issued by inline assembly (invisible to the compiler's wait insertion), flush_pending counts
// the store instructions it issues, and vm_wait_loads waits with exactly that k.
// Where the stores are issued matters as well (same box, LZ4, 64 frames): at the end of the frame's processing 0.428 ms;
// at the top of the next frame, in front of its loads 0.478 ms; right behind those loads 0.482 ms - next to a burst of loads
// they cost more than spread out in time.
struct Pending {
    bool valid;
    uint64_t ft;        // frame * ntiles + tile
    uint32_t f;
    uint32_t cnt;       // residuals staged in LDS
    uint32_t csize;     // LZ4 payload bytes staged in LDS (>= n_blk: store raw); zstd: the blk_size word
    uint32_t staged;    // zstd: bytes staged in LDS for the slot
    u32x2 own;          // this lane's 8 bitmap bytes (raw-block fallback / raw bitmap store)
    uint64_t cown;      // blosc: this lane's 8 bytes of the bit-shuffled block (stored-block fallback)
    uint32_t aux;       // deflate: the tile's Adler-32 partials (rc_deflate_block.h::deflate_adler_word)
    bool last;          // zstd: the tile is the frame's last block
    uint32_t depth;     // bits per staged value (16 = plain uint16)
    uint16_t *buf;      // where the compacted (and packed) values sit in the wave's LDS stage
};

// ---- explicit vmcnt management (see above) -----------------------------------------------------------------------------
// The eight 16-byte loads of one tile of one frame: base + lane*16 + r*1024.  Issued by inline assembly; the results may be
// read only behind vm_wait_loads.
__device__ __forceinline__ void vm_issue_loads(u32x4 (&x)[R], const uint16_t *lane_ptr)
{
    const uint8_t *p0 = reinterpret_cast<const uint8_t *>(lane_ptr), *p1 = p0 + 4096;
    asm volatile("global_load_dwordx4 %0, %4, off nt\n\t"
                 "global_load_dwordx4 %1, %4, off offset:1024 nt\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 nt\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072 nt"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(p0) : "memory");
    asm volatile("global_load_dwordx4 %0, %4, off nt\n\t"
                 "global_load_dwordx4 %1, %4, off offset:1024 nt\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 nt\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072 nt"
                 : "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]) : "v"(p1) : "memory");
}
// uint8 frames: a tile is 4 KiB, a group 512 bytes, a lane's 8 pixels one 8-byte load
__device__ __forceinline__ void vm_issue_loads(u32x2 (&x)[R], const uint8_t *lane_ptr)
{
    asm volatile("global_load_dwordx2 %0, %8, off nt\n\t"
                 "global_load_dwordx2 %1, %8, off offset:512 nt\n\t"
                 "global_load_dwordx2 %2, %8, off offset:1024 nt\n\t"
                 "global_load_dwordx2 %3, %8, off offset:1536 nt\n\t"
                 "global_load_dwordx2 %4, %8, off offset:2048 nt\n\t"
                 "global_load_dwordx2 %5, %8, off offset:2560 nt\n\t"
                 "global_load_dwordx2 %6, %8, off offset:3072 nt\n\t"
                 "global_load_dwordx2 %7, %8, off offset:3584 nt"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]) : "v"(lane_ptr) : "memory");
}
// one of the eight (group r): the rolling form re-arms a group's registers as soon as the group has been consumed
__device__ __forceinline__ void vm_issue_load1(u32x4 &xr, const uint16_t *lane_ptr, int r)
{
    const uint8_t *p = reinterpret_cast<const uint8_t *>(lane_ptr) + (r >= 4 ? 4096 : 0);
    switch (r & 3) {
    case 0: asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 1: asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 2: asm volatile("global_load_dwordx4 %0, %1, off offset:2048 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    default: asm volatile("global_load_dwordx4 %0, %1, off offset:3072 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    }
}
__device__ __forceinline__ void vm_issue_load1(u32x2 &xr, const uint8_t *p, int r)
{
    switch (r) {
    case 0: asm volatile("global_load_dwordx2 %0, %1, off nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 1: asm volatile("global_load_dwordx2 %0, %1, off offset:512 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 2: asm volatile("global_load_dwordx2 %0, %1, off offset:1024 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 3: asm volatile("global_load_dwordx2 %0, %1, off offset:1536 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 4: asm volatile("global_load_dwordx2 %0, %1, off offset:2048 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 5: asm volatile("global_load_dwordx2 %0, %1, off offset:2560 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    case 6: asm volatile("global_load_dwordx2 %0, %1, off offset:3072 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    default: asm volatile("global_load_dwordx2 %0, %1, off offset:3584 nt" : "=&v"(xr) : "v"(p) : "memory"); break;
    }
}
// Wait until at most `later` vector-memory instructions are outstanding, `later` (wave-uniform) being the number issued
// BEHIND the loads of x; an over-estimate would let the loads through unfinished, so anything unusual waits for everything.
// The empty statement at the end names the registers as operands: no use of them can be scheduled in front of the wait.
template <class X>
__device__ __forceinline__ void vm_wait_loads(uint32_t later, X (&x)[R])
{
    switch (later) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) :: "memory");
}

// Wave-private LDS stage of the residual path.  `val` receives the tile's 4096 values in pixel order (8 x ds_write_b128 per
// lane); sparse tiles are compacted from there into `out`, tiles with more than STAGE_CAP set pixels are compacted inside `val`,
// group by group (compact_dense_in_place).  STAGE_CAP = 256 (6.25 % of a tile) is what lets FIVE workgroups of
// three waves share a CU's 160 KB: 3 x (this + Lz4Lds) = 31.9 KB.
#ifndef RC_STAGE_CAP
#define RC_STAGE_CAP 256
#endif
constexpr int STAGE_CAP = RC_STAGE_CAP;   // (experiments: a smaller stage - 16 waves per CU)
struct __attribute__((aligned(16))) WaveStage {
    uint16_t val[TILE_PX];
    uint16_t out[STAGE_CAP];
};
static_assert(offsetof(WaveStage, out) == sizeof(uint16_t) * TILE_PX && STAGE_CAP >= 64, "compact_dense_in_place: `out` lies right behind `val`, an entry per lane");

// All line-sized results of one (tile, frame) - the residual stream's lines and the encoded block's lines, both complete images
// in the wave's LDS - leave through ONE kind of store: every lane moves 16 bytes (ds_read_b128 + global_store_dwordx4), the
// first q0 lanes from segment 0, the next q1 from segment 1, each lane with its own address; 1 KiB per instruction, i.e. one
// instruction for the usual 1 + 2 lines.  Only whole 128-byte lines are written (the tails are unused slot space: partial-line
// writes cost a read-modify-write at the memory side).  The two 4-byte results (count, block size) share one instruction as
// well (lanes 0 and 1, different arrays).  Fewer vector-memory instructions matter here: under the saturated read stream every
// one of them waits for a slot in the CU's memory queue (profiles/r02_reduce_stores.md).
// Returns the number of vector store instructions issued - NEVER more than were issued (vm_wait_loads).
__device__ __forceinline__ uint32_t store_lines2(const uint8_t *s0, uint8_t *d0, uint32_t q0, const uint8_t *s1, uint8_t *d1, uint32_t q1)
{
    const uint32_t total = q0 + q1, lane = (uint32_t)lane_id();
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t i = base + lane;
        if (i < total) {
            const bool second = i >= q0;
            const uint32_t j = second ? i - q0 : i;
            const u32x4 v = *reinterpret_cast<const u32x4 *>((second ? s1 : s0) + 16 * j);
            *reinterpret_cast<u32x4 *>((second ? d1 : d0) + 16 * j) = v;
        }
    }
    return (total + 63u) >> 6;
}

template <bool LEVEL1, int CODEC, bool KEEP_BITMAP>
__device__ __forceinline__ uint32_t flush_pending(const Pending &p, uint32_t tile, uint32_t n_blk, uint8_t *__restrict__ bitmap,
                                                  uint64_t nb_stride, uint16_t *__restrict__ pix_slots, uint32_t *__restrict__ tile_cnt,
                                                  uint8_t *__restrict__ blk_slots, uint32_t *__restrict__ blk_size, Lz4Lds *lz,
                                                  const WaveStage *st, uint32_t blk_stride, uint32_t comb, uint32_t *__restrict__ blk_aux)
{
    if (!p.valid) return 0;
    const int lane = lane_id();
    uint32_t nst = 0;
    // segment 0: the residual lines; segment 1: the encoded block's lines
    const uint8_t *s0 = nullptr, *s1 = nullptr;
    uint8_t *d0 = nullptr, *d1 = nullptr;
    uint32_t q0 = 0, q1 = 0, bsz = 0;
    if (LEVEL1) {
        s0 = reinterpret_cast<const uint8_t *>(p.buf);
        d0 = reinterpret_cast<uint8_t *>(pix_slots + p.ft * SLOT_PX);   // slots are 8 KiB aligned
        q0 = ((((p.cnt * p.depth + 31) >> 5) + 31u) & ~31u) >> 2;       // whole lines, in 16-byte units
    }
    if (CODEC == 2 || CODEC == 4 || CODEC == 8) {
        const uint64_t bytes = CODEC != 8 ? ((uint64_t)p.own[0] | ((uint64_t)p.own[1] << 32)) : p.cown;
        bsz = lz4_stage_slot(bytes, n_blk, p.csize, *lz, CODEC == 8);
    }
    if (CODEC == 1 || CODEC == 3) {
        (void)zstd_stage_slot(n_blk, p.last, p.staged, *lz);
        bsz = p.staged ? p.staged : 4u;
    }
    if (CODEC == 5) bsz = p.csize;   // deflate: the tile's share of the stream stands complete in the stage (rc_deflate_block.h)
    if (CODEC) {
        s1 = lz->out;
        d1 = blk_slots + p.ft * blk_stride;
        q1 = min((((bsz + 3) >> 2) + 31u) & ~31u, (uint32_t)BLK_SLOT / 4) >> 2;
    }
    if (LEVEL1 && CODEC && comb) {
        // combined slot (Scratch::comb, rc_launch.h::residual_src is the reader's side of this rule): the residual stream behind the
        // block image, in the block's slot, when both fit it
        const uint32_t ro16 = comb == 2 ? (uint32_t)BLK_SLOT / 16 : (bsz + 15) >> 4, r16 = (p.cnt * p.depth + 127) >> 7;
        if (16 * (ro16 + r16) <= blk_stride) {
            d0 = d1 + 16 * ro16;
            if (comb == 2) q0 = (r16 + 7u) & ~7u;                                   // two runs of whole lines in one slot
            else { q1 = ro16; q0 = ((ro16 + r16 + 7u) & ~7u) - ro16; }              // ONE run: block units, then residual units up to the line's end
        }
    }
    if (!(RC_ABLATE & 1) && !(RC_ABLATE & 2)) { if (q0 + q1) nst += store_lines2(s0, d0, q0, s1, d1, q1); }
    else if (!(RC_ABLATE & 1)) { if (q0) nst += store_lines2(s0, d0, q0, s1, d1, 0); }
    else if (!(RC_ABLATE & 2)) { if (q1) nst += store_lines2(s1, d1, q1, s0, d0, 0); }
    if (KEEP_BITMAP) {
        *reinterpret_cast<u32x2 *>(bitmap + (uint64_t)p.f * nb_stride + (uint64_t)tile * TILE_BM + lane * 8) = p.own;
        nst += 1;
    }
    // the 4-byte results: lane 0 the count, lane 1 the block's size word, one instruction
    if ((LEVEL1 || CODEC) && !(RC_ABLATE & 4)) {
        const uint32_t word = CODEC == 1 || CODEC == 3 ? p.csize : bsz;   // zstd: the tokenizer's word (k_zstd_fse finishes the block)
        // (deflate: one lane more, the tile's Adler-32 partials)
        if (LEVEL1 && CODEC == 5) {
            if (lane < 3) *(lane == 0 ? &tile_cnt[p.ft] : (lane == 1 ? &blk_size[p.ft] : &blk_aux[p.ft])) = lane == 0 ? p.cnt : (lane == 1 ? word : p.aux);
        } else if (LEVEL1 && CODEC) {
            if (lane < 2) *(lane == 0 ? &tile_cnt[p.ft] : &blk_size[p.ft]) = lane == 0 ? p.cnt : word;
        } else if (LEVEL1) {
            if (lane == 0) tile_cnt[p.ft] = p.cnt;
        } else if (CODEC == 5) {
            if (lane < 2) *(lane == 0 ? &blk_size[p.ft] : &blk_aux[p.ft]) = lane == 0 ? word : p.aux;
        } else {
            if (lane == 0) blk_size[p.ft] = word;
        }
        nst += 1;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)nst);
}

// A tile with more set pixels than `out` holds (STAGE_CAP: 6 % of the tile): compacted INSIDE `val`, group by group.  `val` has the tile's
// values in pixel order and `bm` its mask bytes as the subtract phase left them: byte r * 64 + lane = the 8 pixels r * 512 + 8 * lane + (0..7),
// whose values are ONE 16-byte LDS read for the lane.  For r = 0 .. 7 in turn: every lane reads its 8 values of group r, then writes the
// set ones to val[base_r + (set pixels of the group in the lanes in front) ...), lowest pixel first.  In place: a value's compact index is
// never above its pixel index, so group r's writes end below 512 (r + 1), where the unread groups begin, and a wave's LDS operations execute
// in the order they were issued - the group's reads are in registers before its writes land.  Three wave scans give all eight groups'
// prefixes (counts packed 10 bits apiece: a group's total is at most 512); no loop whose trip count depends on the data, no branch.
// (Until late in round 5 a lane moved the set pixels of ITS 64 consecutive pixels two per trip through a window of STAGE_CAP values, window
// after window: 57 % of a wave's time at 10 % of the pixels set, 88 % at 30 % - profiles/r05_exp7_phase_shares_batch_gaps.log.)
__device__ __forceinline__ void compact_dense_in_place(WaveStage *st, const uint8_t *bm)
{
    const int lane = lane_id();
    uint32_t inc[3], tot[3];     // (the mask bytes are read again group by group: eight more registers would not fit this kernel's 128)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        uint32_t pk = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (3 * g + k < R) pk |= (uint32_t)__builtin_popcount((uint32_t)bm[(3 * g + k) * 64 + lane]) << (10 * k);
        inc[g] = wave_incl_scan(pk);
        tot[g] = wave_last(inc[g]);
    }
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t m = bm[r * 64 + lane];
        const uint32_t dst = base + ((inc[r / 3] >> (10 * (r % 3))) & 0x3FFu) - (uint32_t)__builtin_popcount(m);
        const u32x4 v = (reinterpret_cast<const u32x4 *>(st->val) + lane)[r * 64];
        __builtin_amdgcn_wave_barrier();
        // eight unconditional writes: a pixel that is not set sends its value to this lane's entry of `out` (right behind `val`; a dense tile
        // does not use it) - no branch, no exec-mask round trip per pixel
        uint16_t *const vo = st->val;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t a = (m >> j) & 1u ? dst + (uint32_t)__builtin_popcount(m & ((1u << j) - 1u)) : (uint32_t)TILE_PX + (uint32_t)lane;
            vo[a] = (uint16_t)(v[j >> 1] >> (16 * (j & 1)));
        }
        __builtin_amdgcn_wave_barrier();
        base += (tot[r / 3] >> (10 * (r % 3))) & 0x3FFu;
    }
}

// A3 + A5 in one step for d = 12 (the detector depth of the one acquisition the reference records, BASELINE cfg 5).  After the transpose
// a lane owns 64 consecutive pixels; its cnt set pixels take the compact indices exc .. exc + cnt - 1, i.e. the stream bits
// [12 exc, 12 (exc + cnt)).  Rounds of FOUR values per lane: the lane's next four set pixels are found by bit scans (no loop-carried
// LDS dependency: all four reads of `val` are in flight together), masked to 12 bits and joined to a 48-bit string that is shifted to
// its place in the stream and ORed into the zeroed stage (LDS atomics; neighbouring lanes share a dword).  The round loop is
// wave-uniform: it runs while ANY lane has values left - two rounds at 5 % density where the value-at-a-time loop (above all its
// serial LDS round trips, then pack_stage's own pass over the compact values) made `compaction + pack` the second longest phase of a
// wave (profiles/r03_reduce_phase_shares.md: 2856 of 9904 ticks at 11520 x 8184, 5 %).  `out`: the wave's 512-byte stage = 341 fields.
constexpr uint32_t FUSED12_CAP = (uint32_t)(STAGE_CAP * 16 / 12);   // 12-bit fields the stage holds
__device__ __forceinline__ void compact_pack12(WaveStage *st, const u32x2 &own, uint32_t exc, uint32_t cnt)
{
    const int lane = lane_id();
    uint32_t *out32 = reinterpret_cast<uint32_t *>(st->out);
    *reinterpret_cast<u32x2 *>(out32 + 2 * lane) = u32x2{0u, 0u};       // 64 x 8 bytes: the whole stage
    __builtin_amdgcn_wave_barrier();
    const uint16_t *mine = st->val + 64 * lane;
    uint64_t q = (uint64_t)own[0] | ((uint64_t)own[1] << 32);
    uint32_t left = cnt, P = 12u * exc;                                  // values still to move, stream bit of the next one
    do {
        uint32_t x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = q ? (uint32_t)__builtin_ctzll(q) : 0u;   // (nothing left: any of the lane's own addresses, value dropped below)
            q &= q - 1;
            x[k] = mine[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = (uint32_t)k < left ? (x[k] & 0xFFFu) : 0u;
        const uint32_t lo = x[0] | (x[1] << 12) | (x[2] << 24), hi = (x[2] >> 8) | (x[3] << 4);
        const uint32_t w = P >> 5, sh = P & 31u;                         // (sh is a multiple of 4)
        const uint64_t a = (uint64_t)lo << sh, b = (uint64_t)hi << sh;
        const uint32_t d0 = (uint32_t)a, d1 = (uint32_t)(a >> 32) | (uint32_t)b, d2 = (uint32_t)(b >> 32);
        if (left) {
            __hip_atomic_fetch_or(&out32[w], d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (d1) __hip_atomic_fetch_or(&out32[w + 1], d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (d2) __hip_atomic_fetch_or(&out32[w + 2], d2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        left = left > 4u ? left - 4u : 0u;
        P += 48u;
    } while (__builtin_amdgcn_ballot_w64(left != 0) != 0);
    __builtin_amdgcn_wave_barrier();
}

// One frame of one tile.  x holds the 8 loaded groups of this lane; each group's registers are re-armed with the NEXT frame's load as
// soon as the group has been consumed (ASMLOAD), so the loads stay one frame ahead with ONE register set.
// t: the wave's threshold tile, in registers for all BZ frames (keeping it in LDS or re-reading it from L2 was no faster).
template <bool ALIGNED, bool ASMLOAD, bool LEVEL1, int CODEC, bool KEEP_BITMAP, bool RAWVAL, int SB>
__device__ __forceinline__ void reduce_one_frame(typename Src<SB>::X (&x)[R], const typename Src<SB>::T *__restrict__ cur, const typename Src<SB>::T *__restrict__ next, bool have_next,
                                                 const u32x4 (&t)[R], uint64_t lane_px0, uint64_t N, bool full,
                                                 uint32_t f, uint32_t tile, uint64_t ft, uint32_t n_blk, uint8_t *__restrict__ bitmap,
                                                 uint64_t nb_stride, uint16_t *__restrict__ pix_slots,
                                                 uint32_t *__restrict__ tile_cnt, uint8_t *__restrict__ blk_slots,
                                                 uint32_t *__restrict__ blk_size, Lz4Lds *s_lz, uint8_t *s_bm, WaveStage *st,
                                                 Pending &pend, uint32_t &stores_behind, const ZmParams &zp, uint32_t blk_stride, uint32_t comb,
                                                 uint32_t *__restrict__ blk_aux)
{
    const int lane = lane_id();
    RC_PHASE_BEGIN
    // x was fetched by vm_issue_loads during the previous frame: wait for those loads, not for the stores issued since
    if (ASMLOAD) vm_wait_loads(stores_behind, x);
    else if (ALIGNED && full) {   // (the partial last tile of a frame / unaligned frames: plain loads, no prefetch)
        typedef typename Src<SB>::X XU __attribute__((aligned(SB)));
        const XU *p = reinterpret_cast<const XU *>(cur + lane_px0);
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = __builtin_nontemporal_load(p + r * (GROUP_PX / 8));
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = load8<ALIGNED, true, SB>(cur, lane_px0 + (uint64_t)r * GROUP_PX, N, 0);
    }
    RC_PHASE(0);
    // residuals (saturating subtract, in place) and the 8-bit mask of this lane's 8 pixels, per group
    uint32_t m8[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const u32x4 tt = t[r];
        u32x4 xv = widen8(x[r]);      // (uint8 frames: the 8 pixels as packed uint16 pairs; uint16 frames: the registers themselves)
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = pk_sub_sat_u16(xv[k], tt[k]);
        // 0 / 1 per pixel (packed min with 1), then four chained 16-bit dot products with the bit weights: pixel j -> bit j
        const uint32_t one = 0x00010001u;
        uint32_t M = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k)
            M = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pk_min_u16(xv[k], one)),
                                       __builtin_bit_cast(u16x2, (uint32_t)((1u << (2 * k)) | (2u << (2 * k + 16)))), M, false);
        m8[r] = M;
        if (LEVEL1) {
            // the group's values in pixel order -> LDS (level 2 keeps the raw frame value: residual + threshold)
            u32x4 v = xv;
            if (RAWVAL) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = pk_add_u16(v[k], tt[k]);
            }
            (reinterpret_cast<u32x4 *>(st->val) + lane)[r * 64] = v;
        }
        // the group is consumed: its registers take the NEXT frame's load, which then flies during the whole compaction + encoding
        // of this frame (the whole tile lies inside the frame in this instantiation: no predication)
        if (ASMLOAD && have_next) vm_issue_load1(x[r], next + lane_px0, r);
    }
    RC_PHASE(1);
    RC_PHASE(2);
    pend.valid = true;
    pend.ft = ft;
    pend.f = f;
    pend.cnt = 0;
    pend.csize = 0;
    pend.buf = st->out;
    if (LEVEL1 || KEEP_BITMAP || CODEC) {
        // transpose through wave-private LDS: byte (r, lane) -> position r*64 + lane; 8 contiguous bytes per lane out
        uint8_t *bm = CODEC ? s_lz->raw : s_bm;
#pragma unroll
        for (int r = 0; r < R; ++r) bm[r * 64 + lane] = (uint8_t)m8[r];
        __builtin_amdgcn_wave_barrier();
        pend.own = *reinterpret_cast<const u32x2 *>(&bm[lane * 8]);
    }
    RC_PHASE(3);
    if (LEVEL1) {
        // After the transpose a lane owns 64 CONSECUTIVE pixels (its 8 bitmap bytes): row-major order is lane order, so one
        // prefix sum of the per-lane popcounts places everything, and each lane moves its own set pixels from `val` to
        // `out`, two per step (both LDS reads in flight together).  [The group-by-group alternative below needs three
        // packed scans and eight divergent loops per frame: 0.10 ms of the kernel at 1 % sparsity, measured against level 3.]
        const uint32_t cnt = (uint32_t)__builtin_popcount(pend.own[0]) + (uint32_t)__builtin_popcount(pend.own[1]);
        const uint32_t inc = wave_incl_scan(cnt);
        const uint32_t wave_total = wave_last(inc);
        bool packed = false;
        if (FUSED12 && pend.depth == 12 && wave_total <= FUSED12_CAP) {
            if (wave_total) compact_pack12(st, pend.own, inc - cnt, cnt);
            packed = true;
        } else if (wave_total <= (uint32_t)STAGE_CAP) {
            // (the two 32-pixel halves one after the other: 32-bit bit scans, half the instructions of a 64-bit loop body)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint16_t *mine = st->val + 64 * lane + 32 * h;
                uint32_t qh = pend.own[h];
                uint32_t o = inc - cnt + (h ? (uint32_t)__builtin_popcount(pend.own[0]) : 0u);
                while (qh) {
                    const uint32_t i0 = (uint32_t)__builtin_ctz(qh);
                    qh &= qh - 1;
                    const bool two = qh != 0;
                    const uint32_t i1 = two ? (uint32_t)__builtin_ctz(qh) : i0;
                    qh &= qh - 1;                       // (0 & anything: stays 0)
                    const uint16_t v0 = mine[i0], v1 = mine[i1];
                    st->out[o] = v0;
                    if (two) st->out[o + 1] = v1;
                    o += 2;
                }
            }
        } else {
            pend.buf = st->val;
            __builtin_amdgcn_wave_barrier();
            compact_dense_in_place(st, CODEC ? s_lz->raw : s_bm);
        }
        __builtin_amdgcn_wave_barrier();
        pend.cnt = wave_total;
        if (!packed && pend.depth < 16 && wave_total) pack_stage(pend.buf, wave_total, pend.depth);
    }
    RC_PHASE(4);
    if (CODEC == 2) {
        const uint64_t bytes = (uint64_t)pend.own[0] | ((uint64_t)pend.own[1] << 32);
        pend.csize = lz4_encode_block<false>(bytes, n_blk, *s_lz);
    }
    if (CODEC == 4) {  // LZ4, compression_level >= 1: the event parser (rc_lz4_block.h)
        const uint64_t bytes = (uint64_t)pend.own[0] | ((uint64_t)pend.own[1] << 32);
        pend.csize = lz4_encode_block<true>(bytes, n_blk, *s_lz);
    }
    if (CODEC == 1) {
        const uint64_t bytes = (uint64_t)pend.own[0] | ((uint64_t)pend.own[1] << 32);
        pend.csize = zstd_tokenize_block(bytes, n_blk, pend.last, *s_lz, pend.staged);
    }
    if (CODEC == 3) {  // zstd, modelled: Huffman-coded literals + tokens for the ctx's fitted tables (rc_zstd_wave.h)
        const uint64_t bytes = (uint64_t)pend.own[0] | ((uint64_t)pend.own[1] << 32);
        pend.csize = zstd_tokenize_block_m(bytes, n_blk, pend.last, *s_lz, pend.staged, zp);
    }
    if (CODEC == 5) {  // deflate (compression_scheme 0 on the device): a fixed-Huffman block per tile + its Adler-32 partials (rc_deflate_block.h)
        const uint64_t bytes = (uint64_t)pend.own[0] | ((uint64_t)pend.own[1] << 32);
        pend.aux = deflate_adler_word(bytes, tile);
        pend.csize = deflate_encode_block<true>(bytes, n_blk, pend.last, *s_lz);
    }
    if (CODEC == 8) {  // blosc1 block: bit-shuffle (typesize 8), then the LZ4 block encoder
        const uint64_t bytes = (uint64_t)pend.own[0] | ((uint64_t)pend.own[1] << 32);
        pend.cown = bitshuffle_block(bytes, n_blk, *s_lz);
        pend.csize = lz4_encode_block(pend.cown, n_blk, *s_lz);
    }
    RC_PHASE(5);
    stores_behind = flush_pending<LEVEL1, CODEC, KEEP_BITMAP>(pend, tile, n_blk, bitmap, nb_stride, pix_slots, tile_cnt, blk_slots, blk_size, s_lz, st, blk_stride, comb, blk_aux);
    pend.valid = false;
    RC_PHASE(6);
}

// Workgroup id -> (tile block, frame group).  A tile block is WAVES consecutive tiles (one per wavefront); a frame group
// is BZ consecutive frames.  The threshold tile is fetched once per workgroup and kept in registers
// for the BZ frames.  The ngroups workgroups that share a tile block get ids that are congruent mod 8 and adjacent within
// that residue class: the dispatcher deals workgroups round-robin over the 8 XCDs, so they meet in ONE XCD's L2 at about
// the same time and the threshold is fetched from HBM once per batch (placement affects speed only, never results).
//
// Per frame and wavefront, with NO barrier and no cross-wave traffic (reduce_one_frame):
//   8 x 16-byte nontemporal loads per lane, issued ONE FRAME AHEAD into the second register set
//   -> saturating subtract (residual and mask in one op) -> 8-bit mask per lane
//   -> [LEVEL1] the 4096 values staged in pixel order in the wave's LDS
//   -> bitmap bytes transposed through wave-private LDS (8 contiguous bytes = the mask of 64 consecutive pixels per lane)
//   -> [LEVEL1] one DPP prefix sum of the per-lane popcounts, each lane moves its set pixels into the compact buffer,
//      [depth < 16] packed in place to the tile-local d-bit stream
//   -> [CODEC 2 / 1 / 8] the 512-byte bitmap block is LZ4-encoded / zstd-tokenized / bit-shuffled + LZ4-encoded in LDS
//      (rc_lz4_block.h, rc_zstd_wave.h)
//   -> all global stores (residuals, encoded block, raw bitmap, counts) go out last, as whole 128-byte lines (flush_pending)
//
// ASMLOAD instantiation (aligned frames, every tile of the launch lies wholly inside the frame): the frames are read through
// vm_issue_loads / vm_wait_loads, see Pending.  The other instantiation (the last, partial tile of a frame; unaligned frames)
// leaves loads and waits to the compiler.  tile0: first tile of this launch.
// Workgroup = RWAVES wavefronts.  The wavefronts of this kernel never talk to each other, so the workgroup size is free
// (measurements in launch_reduce_t).
// RWAVES = 3 where the kernel's LDS then lets five workgroups (15 waves) share a CU and the step gains from it, 4 elsewhere (launch_reduce_t)
template <int RWAVES, int BZ, bool ALIGNED, bool ASMLOAD, bool LEVEL1, int CODEC, bool KEEP_BITMAP, bool RAWVAL, int SB>
// (four waves per SIMD = 128 VGPRs, which the steady-state instantiation fits with its one frame register set; the plain-load ones -
// a frame's partial last tile, N % 8 != 0, a frame pointer that is not 16-byte aligned - take what they need)
__global__ __launch_bounds__(64 * RWAVES) __attribute__((amdgpu_waves_per_eu((ALIGNED && ASMLOAD) ? ((RWAVES == 4 && LEVEL1) ? 3 : 4) : 1))) void k_reduce_tiles(const typename Src<SB>::T *__restrict__ frames,
                                                       const uint16_t *__restrict__ thr, uint64_t N, uint32_t ntiles,
                                                       uint32_t tile0, uint32_t tile_end,
                                                       uint32_t B, uint32_t ngroups, uint64_t nb,
                                                       uint8_t *__restrict__ bitmap, uint64_t nb_stride,
                                                       uint16_t *__restrict__ pix_slots, uint32_t *__restrict__ tile_cnt,
                                                       uint8_t *__restrict__ blk_slots, uint32_t *__restrict__ blk_size, uint32_t depth,
                                                       BatchStatus *__restrict__ status, ZmParams zm, uint32_t blk_stride, uint32_t comb,
                                                       uint32_t *__restrict__ blk_aux)
{
#ifdef RC_REDUCE_PRIO
    __builtin_amdgcn_s_setprio(RC_REDUCE_PRIO);   // (experiment: the reduce kernel's waves in front of the second stage's at the issue arbiter)
#endif
    // first kernel of a batch: clears the batch's status word (written later by k_layout / the level-2 kernels)
    if (blockIdx.x == 0 && threadIdx.x == 0) { status->code = 0; status->frame = 0; status->total = 0; }
    __shared__ uint16_t s_code[CODEC == 3 ? 256 : 2];                                           // modelled zstd: Huffman code table
    if (CODEC == 3) {   // (the only barrier of the kernel, in front of every early exit)
        for (uint32_t i = threadIdx.x; i < 128; i += 64 * RWAVES) reinterpret_cast<uint32_t *>(s_code)[i] = reinterpret_cast<const uint32_t *>(zm.lit_code)[i];
        __syncthreads();
        zm.lit_code = s_code;
    }
    __shared__ Lz4Lds s_lz[CODEC ? RWAVES : 1];                                                  // codec working set
    __shared__ __attribute__((aligned(16))) uint8_t s_bm[CODEC ? 1 : RWAVES][CODEC ? 16 : TILE_BM];  // transpose only
    __shared__ WaveStage s_stage[LEVEL1 ? RWAVES : 1];
                                          // compacted residuals

    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t grp = j % ngroups;
    const uint32_t tblock = (j / ngroups) * 8u + xcd;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t tile = tile0 + tblock * RWAVES + w;
    if (tile >= tile_end) return;  // whole wavefront leaves; nothing below synchronises across wavefronts
    const uint64_t lane_px0 = (uint64_t)tile * TILE_PX + (uint64_t)lane * 8;
    const uint32_t f0 = grp * BZ;
    if (f0 >= B) return;
    const bool full = (uint64_t)(tile + 1) * TILE_PX <= N;  // wave-uniform (always true in the ASMLOAD instantiation)

    typename Src<SB>::X xa[R];
    if (ASMLOAD) vm_issue_loads(xa, frames + (uint64_t)f0 * N + lane_px0);
    u32x4 t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = load8<ALIGNED, false, 2>(thr, lane_px0 + (uint64_t)r * GROUP_PX, N, 0xFFFF);
    const uint32_t n_blk = (uint32_t)min((uint64_t)TILE_BM, nb - (uint64_t)tile * TILE_BM);  // bitmap bytes of this tile
    Lz4Lds *lz = &s_lz[CODEC ? w : 0];
    uint8_t *bm = s_bm[CODEC ? 0 : w];
    WaveStage *st = &s_stage[LEVEL1 ? w : 0];
    Pending pend;
    pend.valid = false;
    pend.ft = 0; pend.f = 0; pend.cnt = 0; pend.csize = 0; pend.own = u32x2{0u, 0u};
    pend.staged = 0; pend.last = tile + 1 == ntiles; pend.cown = 0; pend.aux = 0;
    pend.depth = (LEVEL1 && !RAWVAL) ? depth : 16u;
    pend.buf = nullptr;

    // (first frame: the threshold loads were issued behind the frame's; their count is not known to be exact once the
    // compiler has had its way with them, so the first wait is for everything)
    uint32_t stores_behind = 0;
#pragma unroll 1
    for (int z = 0; z < BZ; ++z) {
        const uint32_t f = f0 + z;
        if (f >= B) break;
        const bool nxt = z + 1 < BZ && f + 1 < B;
        reduce_one_frame<ALIGNED, ASMLOAD, LEVEL1, CODEC, KEEP_BITMAP, RAWVAL, SB>(xa, frames + (uint64_t)f * N, frames + (uint64_t)(f + 1) * N, nxt, t,
                                                                             lane_px0, N, full, f, tile, (uint64_t)f * ntiles + tile, n_blk,
                                                                             bitmap, nb_stride, pix_slots, tile_cnt, blk_slots,
                                                                             blk_size, lz, bm, st, pend, stores_behind, zm, blk_stride, comb, blk_aux);
        if (!nxt) break;
    }
    flush_pending<LEVEL1, CODEC, KEEP_BITMAP>(pend, tile, n_blk, bitmap, nb_stride, pix_slots, tile_cnt, blk_slots, blk_size, lz, st, blk_stride, comb, blk_aux);
}

#ifndef RC_RW_ALT
#define RC_RW_ALT 3   // waves per workgroup of the smaller form (experiments: 2)
#endif
template <int BZ, bool AL, bool L1, int CODEC, bool KEEP, bool RAW, int SB>
static void launch_reduce_t(const Scratch &sc, const typename Src<SB>::T *frames, uint32_t B, uint32_t depth, hipStream_t s, hipStream_t s_tail)
{
    const uint32_t ngroups = (B + BZ - 1) / BZ;
    // Workgroup size, measured (bench.py, pipelined, same box; tools/build_def.sh + tools/ab_bench.sh).  With ONE wavefront per
    // workgroup a retiring wavefront frees exactly what the next one needs on its SIMD, and the kernel no longer slows down next to
    // the second stage of the previous batch (11520x8184 5 %: 0.67 ms instead of 0.81; LZ4 1 %: 0.43 instead of 0.46) - because the
    // second stage, whose workgroups need several free slots of one CU at once, is then starved until the reduce kernel has
    // drained: the step gets LONGER (0.97 ms vs 0.83; LZ4 0.483 vs 0.478).  Shrinking the second stage's workgroups to one
    // wavefront as well brings the interference back in full (0.465 ms, step 0.478).  Two wavefronts: in between.  The work is
    // conserved; four wavefronts per workgroup overlaps it best.
    // aligned frames: the tiles that lie wholly inside the frame go through the explicit-wait instantiation; a partial last
    // tile (N not a multiple of TILE_PX) gets a second, tiny launch of the plain one
    const uint32_t nfull = AL ? (uint32_t)(sc.N / TILE_PX) : 0u;
    const ZmParams zm{reinterpret_cast<const uint16_t *>(sc.zm_lit_code), sc.zm_valid, sc.zm_budget, sc.zm_seq_bits};
    const uint32_t comb = (L1 && !RAW && CODEC) ? sc.comb : 0u;   // (level 2 keeps raw values in pix_slots: rc_l2.hip reads them there)
    // Workgroups of THREE waves let five of them (15 waves) share a CU's LDS where four-wave workgroups fit three (12 waves).  Same-box
    // A/B against the two-register-set kernel of round 2 (tools/ab_configs.sh): LZ4 level 1 +1.3..3.5 %, d = 12 +4.5 %, 11520 x 8184
    // zstd +4.5 %; level 3 / mode 0 (nothing to gain from LDS, three-wave workgroups cost 2-4 %) and the configurations whose second
    // stage is long next to the following batch's reduce kernel (zstd at 4096^2, level 2, blosc: the STEP got 1-6 % longer although the
    // kernel got 5 % shorter) keep four-wave workgroups.
    static const char *rw_env = RC_KNOB("RC_REDUCE_WG_WAVES");   // (experiments: 3 or 4)
    // (modelled zstd whose blocks carry literals only - dense maps, rc_zstd_model.h - has no FSE pass behind the reduce kernel: its
    // second stage is as short as LZ4's, and three-wave workgroups gain 0.5-3 % there as well)
    const bool three = rw_env ? atoi(rw_env) == 3 : (L1 && !RAW && CODEC != 5 && (CODEC == 2 || CODEC == 4 || sc.ntiles > 8192 || (CODEC == 3 && (sc.zm_valid & ZM_LITS_ONLY))));
    auto go = [&](auto rw) {
        constexpr int RW = decltype(rw)::value;
        auto grid_for = [&](uint32_t nt) { return (((nt + RW - 1) / RW + 7) / 8) * 8 * ngroups; };
        if (nfull)
            hipLaunchKernelGGL((k_reduce_tiles<RW, BZ, AL, AL, L1, CODEC, KEEP, RAW, SB>), dim3(grid_for(nfull)), dim3(64 * RW), 0, s, frames, sc.thr, sc.N,
                               sc.ntiles, 0u, nfull, B, ngroups, sc.nb, sc.bitmap, sc.nb_stride, sc.pix_slots, sc.tile_cnt, sc.blk_slots,
                               sc.blk_size, depth, sc.status, zm, sc.blk_stride, comb, sc.blk_aux);
        if (nfull < sc.ntiles) {   // (on s_tail: a few workgroups that need not hold up the stream the big launch runs on)
            // the partial last tile: guarded single loads when the frame does not end on a bitmap byte (N % 8 != 0: its last group of
            // eight pixels reaches past the frame), the plain vector-load instantiation otherwise
            if (AL && sc.N % 8 != 0)
                hipLaunchKernelGGL((k_reduce_tiles<RW, BZ, false, false, L1, CODEC, KEEP, RAW, SB>), dim3(grid_for(sc.ntiles - nfull)), dim3(64 * RW), 0, nfull ? s_tail : s, frames,
                                   sc.thr, sc.N, sc.ntiles, nfull, sc.ntiles, B, ngroups, sc.nb, sc.bitmap, sc.nb_stride, sc.pix_slots, sc.tile_cnt,
                                   sc.blk_slots, sc.blk_size, depth, sc.status, zm, sc.blk_stride, comb, sc.blk_aux);
            else
                hipLaunchKernelGGL((k_reduce_tiles<RW, BZ, AL, false, L1, CODEC, KEEP, RAW, SB>), dim3(grid_for(sc.ntiles - nfull)), dim3(64 * RW), 0, nfull ? s_tail : s, frames,
                                   sc.thr, sc.N, sc.ntiles, nfull, sc.ntiles, B, ngroups, sc.nb, sc.bitmap, sc.nb_stride, sc.pix_slots, sc.tile_cnt,
                                   sc.blk_slots, sc.blk_size, depth, sc.status, zm, sc.blk_stride, comb, sc.blk_aux);
        }
    };
    if constexpr (SB == 2) {
        if (three) { go(std::integral_constant<int, RC_RW_ALT>{}); return; }
    }
    go(std::integral_constant<int, 4>{});   // (uint8 frames: four-wave workgroups only - half the instantiations)
}
template <int BZ, bool AL, bool L1, bool RAW, int SB>
static void launch_reduce_c(const Scratch &sc, const typename Src<SB>::T *frames, uint32_t B, uint32_t codec, bool keep, uint32_t depth, hipStream_t s, hipStream_t s_tail)
{
    // raw-value (level 2) instantiations always keep the bitmap: the labelling kernels read it
    if (RAW) keep = true;
#define RC_CODEC(C)                                                                                          \
    do {                                                                                                     \
        if (keep) launch_reduce_t<BZ, AL, L1, C, true, RAW, SB>(sc, frames, B, depth, s, s_tail);                \
        else if (!RAW && C != 0) launch_reduce_t<BZ, AL, L1, C, false, false, SB>(sc, frames, B, depth, s, s_tail); \
    } while (0)
#ifdef RC_DEV_ONLY_CODEC   // (development: one codec's instantiations only - a translation unit that compiles in a quarter of the time, for reading its ISA)
    if (codec == RC_DEV_ONLY_CODEC) RC_CODEC(RC_DEV_ONLY_CODEC);
#else
    if (codec == 2) RC_CODEC(2);
    else if (codec == 4) RC_CODEC(4);
    else if (codec == 1) RC_CODEC(1);
    else if (codec == 3) RC_CODEC(3);
    else if (codec == 8) RC_CODEC(8);
    else if (codec == 5) RC_CODEC(5);
    else launch_reduce_t<BZ, AL, L1, 0, true, RAW, SB>(sc, frames, B, depth, s, s_tail);
#endif
#undef RC_CODEC
}
template <int BZ, bool AL, int SB>
static void launch_reduce_a(const Scratch &sc, const typename Src<SB>::T *frames, uint32_t B, uint32_t level, uint32_t codec, bool keep,
                            uint32_t depth, hipStream_t s, hipStream_t s_tail)
{
    if (level == 2) launch_reduce_c<BZ, AL, true, true, SB>(sc, frames, B, codec, keep, depth, s, s_tail);
    else if (level == 1) launch_reduce_c<BZ, AL, true, false, SB>(sc, frames, B, codec, keep, depth, s, s_tail);
    else launch_reduce_c<BZ, AL, false, false, SB>(sc, frames, B, codec, keep, depth, s, s_tail);
}
#ifdef RC_PHASE_TIMING
extern "C" __attribute__((visibility("default"))) int rc_debug_phases(unsigned long long *out16)
{
    unsigned long long z[16] = {};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof z) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
// s_tail (optional): the stream for the small launch over a frame's partial last tile; it must already be ordered behind
// whatever produced the frames
void launch_reduce(const Scratch &sc, const void *frames, uint32_t B, uint32_t level, uint32_t codec, bool keep_bitmap,
                   uint32_t depth, hipStream_t s, hipStream_t s_tail, uint32_t src_bytes)
{
    if (depth == 0 || depth > 16) depth = 16;
    if (!s_tail) s_tail = s;
    // Vector loads (16 bytes per lane; 8 for uint8 frames) for every tile that lies wholly inside its frame, whatever the frame's
    // alignment: amdhsa runs gfx9+ in unaligned access mode (the compiler emits global_load_dwordx4 for an align-1 16-byte load
    // itself), so a frame may start on any pixel.  Until late in round 4 the rule here was N % 8 == 0 and a 16-byte aligned base, and
    // 3838 x 3710 frames - a common detector format, N % 8 = 4 - took the guarded single loads throughout, at a sixth of the rate.
    // (Scratch::guarded_loads - RC_REDUCE_GUARDED_LOADS=1 when the ctx was created: the guarded instantiation for every tile - tests keep it alive.)
    const bool aligned = !sc.guarded_loads;
    if (src_bytes == 1) {   // uint8 frames (source_bit_depth <= 8)
        const uint8_t *f8 = static_cast<const uint8_t *>(frames);
        if (aligned) launch_reduce_a<RC_BZ, true, 1>(sc, f8, B, level, codec, keep_bitmap, depth, s, s_tail);
        else launch_reduce_a<RC_BZ, false, 1>(sc, f8, B, level, codec, keep_bitmap, depth, s, s_tail);
        return;
    }
    const uint16_t *f16 = static_cast<const uint16_t *>(frames);
    if (aligned) launch_reduce_a<RC_BZ, true, 2>(sc, f16, B, level, codec, keep_bitmap, depth, s, s_tail);
    else launch_reduce_a<RC_BZ, false, 2>(sc, f16, B, level, codec, keep_bitmap, depth, s, s_tail);
}

// ---- per-frame scans over tiles ---------------------------------------------------------------------------
// One 256-thread workgroup per frame, SCAN_I consecutive entries per thread and round (4096 entries per round):
//   tile_off = exclusive prefix of tile_cnt, frame_nnz = its total, tile_next[t] = smallest t' > t with tile_cnt[t'] > 0
//   blk_off  = exclusive prefix of blk_size, frame_cbytes = its total              (only when a device codec ran)
// The workgroup is deliberately small: in pipelined mode this kernel is dispatched while the next batch's reduce kernel
// owns the chip, and a 1024-thread workgroup (16 waves that must start on one CU together) waited there for hundreds of
// microseconds (rocprofv3: 74 us with LZ4, 355 us with zstd, against 10 us when alone).
#ifndef RC_SCAN_T
#define RC_SCAN_T 256
#endif
constexpr int SCAN_T = RC_SCAN_T, SCAN_W = SCAN_T / 64;
// entries per thread and round: 16 (one round for 4096 tiles), 32 for frames with many more tiles (half the rounds, each of which is a
// dependent global round trip + two barriers: 11520x8184 has 23 018 tiles)

__device__ __forceinline__ uint32_t scan_block_excl(uint32_t v, uint32_t *sm, uint32_t *total)
{
    const int w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);
    if (lane_id() == 63) sm[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_W; ++i) {
        const uint32_t x = sm[i];
        if (i < w) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// adj (optional): {tile a, bytes added to a, tile b, bytes added to b}: the row's size words are read without their flag bits,
// the two additions applied, and the clean sizes written back
template <int SCAN_I>
__device__ __forceinline__ void scan_row(uint32_t *__restrict__ row, uint32_t *__restrict__ orow, uint32_t n,
                                         uint32_t *__restrict__ total, uint32_t *sm, const uint32_t *adj = nullptr)
{
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += SCAN_T * SCAN_I) {
        const uint32_t t = t0 + threadIdx.x * SCAN_I;
        uint32_t v[SCAN_I], s = 0;
#pragma unroll
        for (int k = 0; k < SCAN_I; ++k) {
            v[k] = t + k < n ? row[t + k] : 0;
            if (adj) {
                v[k] &= 0xFFFFu;
                if (t + k == adj[0]) v[k] += adj[1];
                if (t + k == adj[2]) v[k] += adj[3];
                if (t + k < n) row[t + k] = v[k];
            }
            s += v[k];
        }
        uint32_t tot;
        uint32_t ex = carry + scan_block_excl(s, sm, &tot);
#pragma unroll
        for (int k = 0; k < SCAN_I; ++k) {
            if (t + k < n) orow[t + k] = ex;
            ex += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

// The frame's definitions go into block t_tree (tree behind the literals header) and block t_seq (table descriptions behind the
// modes byte) - usually the same block: the workgroup copies the block's slot into LDS and rewrites it byte by byte through
// zm_defs_byte (the block encoders left room).  which: bit 0 = do t_tree's block, bit 1 = do t_seq's (when it is another block).
__device__ __forceinline__ void zstd_rewrite_defs(const uint32_t *__restrict__ row, uint8_t *__restrict__ slots, uint32_t stride,
                                                  const uint8_t *__restrict__ tree, uint32_t tl, const uint8_t *__restrict__ sdesc, uint32_t sl,
                                                  uint32_t t_tree, uint32_t t_seq, uint32_t which, uint8_t *s_img, uint32_t *s_pos)
{
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t t = pass == 0 ? t_tree : t_seq;
        if (t == 0xFFFFFFFFu || (pass == 1 && t == t_tree) || !((which >> pass) & 1u)) continue;   // (uniform: shared values)
        const uint32_t a_tl = pass == 0 ? tl : 0u, a_sl = (pass == 1 || t_seq == t_tree) ? sl : 0u;
        uint8_t *slot = slots + (uint64_t)t * stride;
        const uint32_t size = row[t] & 0xFFFFu;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (size + 3) / 4; i += blockDim.x)
            reinterpret_cast<uint32_t *>(s_img)[i] = reinterpret_cast<const uint32_t *>(slot)[i];
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t q; (void)zm_block_needs(s_img, &q); *s_pos = q; }
        __syncthreads();
        const uint32_t pos = *s_pos;
        for (uint32_t i = threadIdx.x; i < size + a_tl + a_sl; i += blockDim.x)
            slot[i] = zm_defs_byte(s_img, pos, tree, a_tl, sdesc, a_sl, i);
    }
    __syncthreads();
}

// Modelled zstd: every block of a frame was encoded as if the decoder already had the frame's Huffman tree and sequence
// tables (rc_zstd_wave.h, rc_pix_huff.hip).  The first block that uses the tree (ZW_TREE in its size word) and the first whose
// sequences use the tables (ZW_SEQ) - usually the same one, the frame's first - get the descriptions inserted here: the
// workgroup copies the block into LDS and rewrites its slot byte by byte through zm_defs_byte (the block encoders left
// room).  row: the frame's size words; adj receives what scan_row has to add to the two blocks' sizes.
__device__ __forceinline__ void zstd_place_defs(const uint32_t *__restrict__ row, uint32_t n, uint8_t *__restrict__ slots, uint32_t stride,
                                                const uint8_t *__restrict__ tree, uint32_t tree_len, const uint8_t *__restrict__ sdesc,
                                                uint32_t sdesc_len, uint32_t *adj)
{
    __shared__ uint32_t s_first[2];
    __shared__ uint32_t s_pos;
    __shared__ __attribute__((aligned(16))) uint8_t s_img[PIX_SLOT + 16];
    if (threadIdx.x < 2) s_first[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t w0 = row[0];
    const bool both0 = (w0 & ZW_TREE) && ((w0 & ZW_SEQ) || !sdesc_len);   // the usual case: the frame's first block takes both
    if (both0) {
        if (threadIdx.x == 0) { s_first[0] = 0; if (sdesc_len) s_first[1] = 0; }
    } else {
        uint32_t mt = 0xFFFFFFFFu, mq = 0xFFFFFFFFu;
        for (uint32_t t = threadIdx.x; t < n; t += SCAN_T) {
            const uint32_t w = row[t];
            if ((w & ZW_TREE) && t < mt) mt = t;
            if ((w & ZW_SEQ) && t < mq) mq = t;
        }
        if (mt != 0xFFFFFFFFu) atomicMin(&s_first[0], mt);
        if (mq != 0xFFFFFFFFu) atomicMin(&s_first[1], mq);
    }
    __syncthreads();
    const uint32_t t_tree = s_first[0], t_seq = s_first[1];
    const uint32_t tl = t_tree != 0xFFFFFFFFu ? tree_len : 0u, sl = t_seq != 0xFFFFFFFFu ? sdesc_len : 0u;
    if (threadIdx.x == 0) { adj[0] = t_tree; adj[1] = tl; adj[2] = t_seq; adj[3] = sl; }
    zstd_rewrite_defs(row, slots, stride, tree, tl, sdesc, sl, t_tree, t_seq, 3u, s_img, &s_pos);
}

// Residual stream of the modelled zstd encoder (rc_pix_huff.hip): per frame, sizes of the encoded chunks -> offsets, total;
// the tree goes into the first Huffman-coded chunk.
__global__ __launch_bounds__(SCAN_T) void k_pix_scan(Scratch sc, uint32_t depth)
{
    __shared__ uint32_t sm[SCAN_W];
    __shared__ uint32_t s_adj[4];
    const uint32_t f = blockIdx.x;
    const ZstdModel *M = reinterpret_cast<const ZstdModel *>(sc.zm_model);
    const uint32_t npk = packed_bytes(sc.frame_nnz[f], depth);
    const uint32_t nch = npk ? (npk + PIX_CHUNK - 1) / PIX_CHUNK : 1u;
    uint32_t *row = sc.chunk_size + (uint64_t)f * sc.nchunk_max;
    zstd_place_defs(row, nch, sc.pix_chunks + (uint64_t)f * sc.nchunk_max * PIX_SLOT, PIX_SLOT, M->pix_desc, M->pix_desc_len, nullptr, 0, s_adj);
    scan_row<16>(row, sc.chunk_off + (uint64_t)f * sc.nchunk_max, nch, sc.frame_pbytes + f, sm, s_adj);
}
void launch_pix_scan(const Scratch &sc, uint32_t B, uint32_t depth, hipStream_t s)
{
    hipLaunchKernelGGL(k_pix_scan, dim3(B), dim3(SCAN_T), 0, s, sc, depth);
}

template <int SCAN_I>
__global__ __launch_bounds__(SCAN_T) void k_scan_frames(Scratch sc, int with_counts, int with_blocks)
{
    __shared__ uint32_t sm[SCAN_W];
    __shared__ uint32_t s_carry;
    __shared__ uint32_t s_adj[4];
    const uint32_t f = blockIdx.x, n = sc.ntiles;
    const uint64_t fr = (uint64_t)f * n;
    if (with_blocks) {
        if (sc.zm_model) {
            const ZstdModel *M = reinterpret_cast<const ZstdModel *>(sc.zm_model);
            zstd_place_defs(sc.blk_size + fr, n, sc.blk_slots + fr * sc.blk_stride, sc.blk_stride, M->lit_desc, M->lit_desc_len, M->seq_desc,
                            M->seq_desc_len, s_adj);
        }
        scan_row<SCAN_I>(sc.blk_size + fr, sc.blk_off + fr, n, sc.frame_cbytes + f, sm, sc.zm_model ? s_adj : nullptr);
    }
    if (!with_counts) return;
    uint32_t *row = sc.tile_cnt + fr;
    scan_row<SCAN_I>(row, sc.tile_off + fr, n, sc.frame_nnz + f, sm);
    // suffix pass, rounds from the end: next non-empty tile
    uint32_t *nrow = sc.tile_next + fr;
    if (threadIdx.x == 0) s_carry = n;
    __syncthreads();
    const uint32_t per_round = SCAN_T * SCAN_I;
    const uint32_t nrounds = (n + per_round - 1) / per_round;
    for (uint32_t c = nrounds; c-- > 0;) {
        const uint32_t t = c * per_round + threadIdx.x * SCAN_I;
        uint32_t v[SCAN_I];
        uint32_t first = 0xFFFFFFFFu;  // smallest non-empty index of this thread's chunk
#pragma unroll
        for (int k = SCAN_I - 1; k >= 0; --k) {
            v[k] = t + k < n ? row[t + k] : 0;
            if (v[k]) first = t + k;
        }
        uint32_t m = first;  // inclusive suffix-min over the lanes of the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_down(m, d);
            if (lane_id() + d < 64) m = min(m, y);
        }
        const int w = threadIdx.x >> 6;
        if (lane_id() == 0) sm[w] = m;
        __syncthreads();
        uint32_t right = s_carry;  // min over all later rounds, then over the later waves of this round
#pragma unroll
        for (int i = SCAN_W - 1; i >= 0; --i)
            if (i > w) right = min(right, sm[i]);
        uint32_t excl = __shfl_down(m, 1);
        if (lane_id() == 63) excl = 0xFFFFFFFFu;
        uint32_t nxt = min(excl, right);  // next non-empty index behind this thread's chunk
#pragma unroll
        for (int k = SCAN_I - 1; k >= 0; --k) {
            if (t + k < n) nrow[t + k] = nxt;
            if (v[k]) nxt = t + k;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t all = s_carry;
#pragma unroll
            for (int i = 0; i < SCAN_W; ++i) all = min(all, sm[i]);
            s_carry = all;
        }
        __syncthreads();
    }
}

// ---- the same scans for frames with many tiles (11520 x 8184: 23 018), cut into segments of SEG tiles ------------------------
// One workgroup per frame walks such a row in dependent rounds (three arrays, a global round trip and two barriers per round:
// 228 us for 32 frames).  Here a workgroup takes ONE segment: k_scan_seg leaves segment-local prefixes, next-non-empty links that
// end at the segment's border, and per segment {block bytes, set pixels, first non-empty tile, first block that needs the tree,
// first that needs the sequence tables}; k_scan_fix reads the frame's handful of partials, places the frame's zstd definitions
// (the workgroup whose segment holds the block), adds the bases and closes the links across the borders.  Same results as
// k_scan_frames, entry for entry.
constexpr int SEG_I = 16, SEG = SCAN_T * SEG_I;   // 4096 tiles per segment
struct ScanPart { uint32_t blk_sum, cnt_sum, first_nonempty, first_tree, first_seq, pad[3]; };

__global__ __launch_bounds__(SCAN_T) void k_scan_seg(Scratch sc, int with_counts, int with_blocks)
{
    __shared__ uint32_t sm[SCAN_W];
    __shared__ uint32_t s_min[3];
    const uint32_t seg = blockIdx.x, f = blockIdx.y, n = sc.ntiles, nseg = gridDim.x;
    const uint64_t fr = (uint64_t)f * n;
    const uint32_t t = seg * SEG + threadIdx.x * SEG_I;
    ScanPart *part = reinterpret_cast<ScanPart *>(sc.scan_part) + (uint64_t)f * nseg + seg;
    if (threadIdx.x < 3) s_min[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t blk_sum = 0, cnt_sum = 0;
    if (with_blocks) {
        uint32_t *row = sc.blk_size + fr, *orow = sc.blk_off + fr;
        const bool zm = sc.zm_model != nullptr;
        uint32_t v[SEG_I], sum = 0, mt = 0xFFFFFFFFu, mq = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < SEG_I; ++k) {
            const uint32_t w = t + k < n ? row[t + k] : 0;
            if (zm) {
                if ((w & ZW_TREE) && mt == 0xFFFFFFFFu) mt = t + k;
                if ((w & ZW_SEQ) && mq == 0xFFFFFFFFu) mq = t + k;
                v[k] = w & 0xFFFFu;
                if (t + k < n) row[t + k] = v[k];        // the clean size (k_scan_fix adds the definitions' bytes to two of them)
            } else v[k] = w;
            sum += v[k];
        }
        if (mt != 0xFFFFFFFFu) atomicMin(&s_min[1], mt);
        if (mq != 0xFFFFFFFFu) atomicMin(&s_min[2], mq);
        uint32_t ex = scan_block_excl(sum, sm, &blk_sum);
#pragma unroll
        for (int k = 0; k < SEG_I; ++k) {
            if (t + k < n) orow[t + k] = ex;
            ex += v[k];
        }
    }
    if (with_counts) {
        const uint32_t *row = sc.tile_cnt + fr;
        uint32_t *orow = sc.tile_off + fr, *nrow = sc.tile_next + fr;
        uint32_t v[SEG_I], sum = 0, first = 0xFFFFFFFFu;
#pragma unroll
        for (int k = SEG_I - 1; k >= 0; --k) {
            v[k] = t + k < n ? row[t + k] : 0;
            if (v[k]) first = t + k;
            sum += v[k];
        }
        uint32_t ex = scan_block_excl(sum, sm, &cnt_sum);
#pragma unroll
        for (int k = 0; k < SEG_I; ++k) {
            if (t + k < n) orow[t + k] = ex;
            ex += v[k];
        }
        // next non-empty tile inside the segment: suffix minimum of the threads' first non-empty index
        uint32_t m = first;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_down(m, d);
            if (lane_id() + d < 64) m = min(m, y);
        }
        const int w = threadIdx.x >> 6;
        __syncthreads();
        if (lane_id() == 0) sm[w] = m;
        __syncthreads();
        uint32_t right = 0xFFFFFFFFu;
#pragma unroll
        for (int i = SCAN_W - 1; i >= 0; --i)
            if (i > w) right = min(right, sm[i]);
        uint32_t excl = __shfl_down(m, 1);
        if (lane_id() == 63) excl = 0xFFFFFFFFu;
        uint32_t nxt = min(excl, right);   // 0xFFFFFFFF: nothing behind this thread's chunk inside the segment (k_scan_fix closes it)
#pragma unroll
        for (int k = SEG_I - 1; k >= 0; --k) {
            if (t + k < n) nrow[t + k] = nxt;
            if (v[k]) nxt = t + k;
        }
        if (first != 0xFFFFFFFFu) atomicMin(&s_min[0], first);
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // (only what this launch computed: the count half and the block half of a batch may run side by side on two streams)
        if (with_blocks) { part->blk_sum = blk_sum; part->first_tree = s_min[1]; part->first_seq = s_min[2]; }
        if (with_counts) { part->cnt_sum = cnt_sum; part->first_nonempty = s_min[0]; }
    }
}

__global__ __launch_bounds__(SCAN_T) void k_scan_fix(Scratch sc, int with_counts, int with_blocks)
{
    __shared__ uint32_t s_v[8];   // blk base, cnt base, next non-empty behind the segment, t_tree, t_seq, blk total, cnt total
    __shared__ uint32_t s_pos;
    __shared__ __attribute__((aligned(16))) uint8_t s_img[PIX_SLOT + 16];
    const uint32_t seg = blockIdx.x, f = blockIdx.y, n = sc.ntiles, nseg = gridDim.x;
    const uint64_t fr = (uint64_t)f * n;
    const ScanPart *parts = reinterpret_cast<const ScanPart *>(sc.scan_part) + (uint64_t)f * nseg;
    if (threadIdx.x == 0) {
        uint32_t bb = 0, cb = 0, bt = 0, ct = 0, nx = n, tt = 0xFFFFFFFFu, tq = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < nseg; ++i) {
            const ScanPart p = parts[i];
            if (i < seg) { bb += p.blk_sum; cb += p.cnt_sum; }
            bt += p.blk_sum; ct += p.cnt_sum;
            if (i > seg && p.first_nonempty < nx) nx = p.first_nonempty;
            tt = min(tt, p.first_tree); tq = min(tq, p.first_seq);
        }
        s_v[0] = bb; s_v[1] = cb; s_v[2] = nx; s_v[3] = tt; s_v[4] = tq; s_v[5] = bt; s_v[6] = ct;
    }
    __syncthreads();
    const uint32_t lo = seg * SEG, hi = min(lo + (uint32_t)SEG, n);
    if (with_blocks) {
        uint32_t tl = 0, sl = 0;
        const uint32_t t_tree = s_v[3], t_seq = s_v[4];
        if (sc.zm_model) {
            const ZstdModel *M = reinterpret_cast<const ZstdModel *>(sc.zm_model);
            tl = t_tree != 0xFFFFFFFFu ? M->lit_desc_len : 0u;
            sl = t_seq != 0xFFFFFFFFu ? M->seq_desc_len : 0u;
            const uint32_t mine = (t_tree >= lo && t_tree < hi ? 1u : 0u) | (t_seq >= lo && t_seq < hi ? 2u : 0u);
            if (mine) {
                zstd_rewrite_defs(sc.blk_size + fr, sc.blk_slots + fr * sc.blk_stride, sc.blk_stride, M->lit_desc, tl, M->seq_desc, sl, t_tree, t_seq, mine, s_img, &s_pos);
                if (threadIdx.x == 0) {   // (behind the rewrite, which reads the blocks' clean sizes)
                    if (mine & 1u) sc.blk_size[fr + t_tree] += tl;
                    if (mine & 2u) sc.blk_size[fr + t_seq] += sl;
                }
            }
        }
        uint32_t *orow = sc.blk_off + fr;
        const uint32_t base = s_v[0];
        for (uint32_t t = lo + threadIdx.x; t < hi; t += SCAN_T)
            orow[t] += base + (t > t_tree ? tl : 0u) + (t > t_seq ? sl : 0u);   // (t_tree, t_seq == 0xFFFFFFFF: never)
        if (seg == 0 && threadIdx.x == 0) sc.frame_cbytes[f] = s_v[5] + tl + sl;
    }
    if (with_counts) {
        uint32_t *orow = sc.tile_off + fr, *nrow = sc.tile_next + fr;
        const uint32_t base = s_v[1], nx = s_v[2];
        for (uint32_t t = lo + threadIdx.x; t < hi; t += SCAN_T) {
            orow[t] += base;
            if (nrow[t] == 0xFFFFFFFFu) nrow[t] = nx;
        }
        if (seg == 0 && threadIdx.x == 0) sc.frame_nnz[f] = s_v[6];
    }
}

void launch_scans(const Scratch &sc, uint32_t B, bool with_counts, bool with_blocks, hipStream_t s)
{
    if (!with_counts && !with_blocks) return;
    if (sc.ntiles > (uint32_t)SEG && sc.scan_part) {   // (a ctx's scratch; the stateless seams scan their one row with the kernel below)
        const dim3 grid((sc.ntiles + SEG - 1) / SEG, B);
        hipLaunchKernelGGL(k_scan_seg, grid, dim3(SCAN_T), 0, s, sc, with_counts ? 1 : 0, with_blocks ? 1 : 0);
        hipLaunchKernelGGL(k_scan_fix, grid, dim3(SCAN_T), 0, s, sc, with_counts ? 1 : 0, with_blocks ? 1 : 0);
        return;
    }
    if (sc.ntiles > 8192) hipLaunchKernelGGL(k_scan_frames<32>, dim3(B), dim3(SCAN_T), 0, s, sc, with_counts ? 1 : 0, with_blocks ? 1 : 0);
    else hipLaunchKernelGGL(k_scan_frames<16>, dim3(B), dim3(SCAN_T), 0, s, sc, with_counts ? 1 : 0, with_blocks ? 1 : 0);
}

// ---- record layout: sizes, offsets, metadata, status (stream framing: rc_record.h) ----------------------------------------
constexpr int LWG = WG;
__global__ __launch_bounds__(LWG) void k_layout(const uint32_t *__restrict__ frame_nnz,
                                                 const uint32_t *__restrict__ frame_cbytes, const uint32_t *__restrict__ frame_pbytes,
                                                 RecordParams rp, uint64_t nb,
                                                 uint32_t ntiles, uint32_t B, uint64_t out_cap, uint64_t *__restrict__ rec_off,
                                                 uint32_t *__restrict__ md, BatchStatus *__restrict__ st, u32x4 *__restrict__ zl_acc)
{
    __shared__ uint64_t s_part[LWG];
    __shared__ uint32_t s_bad;
    if (threadIdx.x == 0) s_bad = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t per = (B + LWG - 1) / LWG;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = min(lo + per, B);
    uint64_t sum = 0;
    for (uint32_t f = lo; f < hi; ++f) {
        const uint32_t nnz = rp.level == 1 ? frame_nnz[f] : 0;
        const uint32_t npk = rp.level == 1 ? packed_bytes(nnz, rp.depth) : 0;
        uint64_t sz;
        uint32_t m0 = 0, m1 = 0, m2 = 0;
        if (rp.emit == 0) {
            if (rp.level == 1) { sz = 8 + nb + npk; m0 = npk; }
            else sz = 4 + nb;
        } else {  // LZ4 frames (emit == 2)
            const FrameFmt ff = frame_fmt(rp.emit);
            const uint32_t cb = bitmap_hdr(ff, rp.emit, ntiles) + frame_cbytes[f] + ff.end;
            if (rp.level == 1) {
                const uint32_t cp = rp.pix_mode == 2 ? ff.hdr + frame_pbytes[f] : stored_size(ff, npk);
                sz = 16 + (uint64_t)cb + cp; m0 = cb; m1 = cp; m2 = npk;
            } else { sz = 8 + (uint64_t)cb; m0 = cb; }
        }
        md[3 * f] = m0; md[3 * f + 1] = m1; md[3 * f + 2] = m2;
        if (zl_acc) { zl_acc[2 * f] = u32x4{0u, 0u, 0u, 0u}; zl_acc[2 * f + 1] = u32x4{0u, 0u, 0u, 0u}; }   // deflate: the frame's Adler-32 sums
        if (sz > rp.frame_bytes) atomicMin(&s_bad, f);
        rec_off[f + 1] = sz;  // sizes first; turned into offsets below
        sum += sz;
    }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t i = 0; i < threadIdx.x; ++i) base += s_part[i];
    for (uint32_t f = lo; f < hi; ++f) {
        const uint64_t sz = rec_off[f + 1];
        base += sz;
        rec_off[f + 1] = base;
    }
    if (threadIdx.x == 0) rec_off[0] = 0;
    __syncthreads();
    if (threadIdx.x == LWG - 1) {
        uint64_t total = 0;
        for (uint32_t i = 0; i < LWG; ++i) total += s_part[i];
        st->total = total;
        if (s_bad != 0xFFFFFFFFu) { st->code = -5; st->frame = s_bad; }        // RC_ERR_RECORD_TOO_LARGE
        else if (total > out_cap) { st->code = -2; st->frame = 0; }            // RC_ERR_OUT_TOO_SMALL
    }
}

void launch_layout(const Scratch &sc, const RecordParams &rp, uint32_t B, uint64_t out_cap, uint64_t *rec_off,
                   uint32_t *md, hipStream_t s)
{
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(LWG), 0, s, sc.frame_nnz, sc.frame_cbytes, sc.frame_pbytes, rp, sc.nb, sc.ntiles, B, out_cap,
                       rec_off, md, sc.status, reinterpret_cast<u32x4 *>(sc.zl_acc));
}

// ---- record assembly: k_gather (rc_gather.hip); here the LZ4 frame descriptors and its launcher ----------------------------------
// xxHash32 of the two descriptor bytes -> LZ4 frame header checksum byte (lz4_Frame_format.md, "HC")
static uint32_t xxh32_small(const uint8_t *p, size_t n)
{
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    (void)P1; (void)P4;
    uint32_t h = 0 /*seed*/ + P5 + (uint32_t)n;
    for (size_t i = 0; i < n; ++i) {
        h += p[i] * P5;
        h = ((h << 11) | (h >> 21)) * P1;
    }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
uint32_t lz4f_descriptor(uint8_t bd)
{
    const uint8_t desc[2] = {0x60 /* version 01, block-independent, no checksums, no content size */, bd};
    const uint8_t hc = (uint8_t)((xxh32_small(desc, 2) >> 8) & 0xFF);
    return desc[0] | ((uint32_t)desc[1] << 8) | ((uint32_t)hc << 16);
}

void launch_assemble(const Scratch &sc, const RecordParams &rp, uint32_t B, uint8_t *out, const uint64_t *rec_off,
                     uint32_t batch_seq, hipStream_t s)
{
    static const uint32_t hdr_bitmap = lz4f_descriptor(0x40);  // 64 KiB max block (blocks are <= 2 KiB)
    static const uint32_t hdr_pix = lz4f_descriptor(0x70);     // 4 MiB max block (stored chunks)
    launch_gather(sc, rp, B, out, rec_off, hdr_bitmap, hdr_pix, batch_seq, s);
}

}  // namespace rc
