// Development probe (not part of the product): what does a small write stream cost the reduce kernel's read stream, and
// where does the time go?  Every kernel reads 2 GiB exactly like k_reduce_tiles does (same grid shape, XCD mapping, 4 frames
// per wave, nontemporal 16-byte loads) and writes WL full 128-byte lines per (tile, frame) in one of several ways.  Each
// variant is its own kernel name so that `rocprofv3 --pmc` rows can be told apart (tools/prof_wr_probe.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void map(uint32_t G, uint32_t &tb, uint32_t &g)
{
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    g = j % G; tb = (j / G) * 8 + xcd;
}
__device__ __forceinline__ u32x4 read_tile(const u32x4 *p, uint64_t frame16, uint32_t f, uint32_t tb, uint32_t w, uint32_t lane)
{
    const u32x4 *fr = p + (uint64_t)f * frame16 + (uint64_t)tb * 2048 + w * 512 + lane;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
    return acc;
}
__device__ __forceinline__ uint32_t fold(const u32x4 &a) { return a[0] ^ a[1] ^ a[2] ^ a[3]; }
#define SINK(x) if ((x) == 0x12345678u) wbuf[0] = 1   /* never true: keeps every lane's loads alive */

__global__ __launch_bounds__(256) void k_pmc_read(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *out)
{
    uint32_t tb, g; map(G, tb, g);
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    for (int z = 0; z < 4; ++z) acc += read_tile(p, frame16, g * 4 + z, tb, w, lane);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

// MODE 0: plain dword stores (what the product does)   1: nontemporal   2: write-through (sc0 sc1)
// MODE 5: 16-byte stores (8 lanes per line)            6: 16-byte stores, sc0 sc1
template <int WL, int MODE>
__global__ __launch_bounds__(256) void k_pmc_rw(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf)
{
    uint32_t tb, g; map(G, tb, g);
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t sink = 0;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        u32x4 acc = read_tile(p, frame16, f, tb, w, lane);
        sink += fold(acc);
        acc[0] = fold(acc);
        const uint64_t slot = (uint64_t)f * ntb * 4 + (uint64_t)tb * 4 + w;
        uint32_t *dst = wbuf + 64 + slot * 160;   // 640-byte slots
        if (MODE == 5 || MODE == 6) {
            if (lane < 8u * WL) {
                u32x4 *d = reinterpret_cast<u32x4 *>(dst) + lane;
                if (MODE == 5) *d = acc;
                else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d), "v"(acc) : "memory");
            }
        } else {
#pragma unroll
            for (int l = 0; l < WL; ++l)
                if (lane < 32) {
                    uint32_t *d = dst + l * 32 + lane;
                    const uint32_t v = acc[0] + l;
                    if (MODE == 0) *d = v;
                    else if (MODE == 1) __builtin_nontemporal_store(v, d);
                    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(d), "v"(v) : "memory");
                }
        }
    }
    SINK(sink);
}

// MODE 3: one writer wave per workgroup: every wave leaves its lines in LDS, after a barrier wave 0 stores all four tiles'
// lines (the four tiles of a workgroup are adjacent slots: 4 * WL lines = one burst from one wave)
template <int WL>
__global__ __launch_bounds__(256) void k_pmc_rw_onewriter(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf)
{
    __shared__ uint32_t s[4][WL * 32];
    uint32_t tb, g; map(G, tb, g);
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t sink = 0;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const u32x4 acc = read_tile(p, frame16, f, tb, w, lane);
        sink += fold(acc);
        for (int l = 0; l < WL; ++l) if (lane < 32) s[w][l * 32 + lane] = fold(acc) + l;
        __syncthreads();
        if (w == 0) {
            uint32_t *dst = wbuf + 64 + ((uint64_t)f * ntb * 4 + (uint64_t)tb * 4) * (WL * 32);   // contiguous 4 * WL lines
            const uint32_t *src = &s[0][0];
            for (uint32_t i = lane; i < 4u * WL * 32; i += 64) dst[i] = src[i];
        }
        __syncthreads();
    }
    SINK(sink);
}

// MODE 4: the whole workgroup's output of its 4 frames (4 waves x 4 frames x WL lines) kept in LDS and written at the end as
// one contiguous burst of 16 * WL lines (>= 2 KiB)
template <int WL>
__global__ __launch_bounds__(256) void k_pmc_rw_wgburst(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf)
{
    __shared__ uint32_t s[4][4][WL * 32];
    uint32_t tb, g; map(G, tb, g);
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t sink = 0;
    for (int z = 0; z < 4; ++z) {
        const u32x4 acc = read_tile(p, frame16, g * 4 + z, tb, w, lane);
        sink += fold(acc);
        for (int l = 0; l < WL; ++l) if (lane < 32) s[z][w][l * 32 + lane] = fold(acc) + l;
    }
    __syncthreads();
    uint32_t *dst = wbuf + 64 + ((uint64_t)tb * G + g) * (16u * WL * 32);
    const uint32_t *src = &s[0][0][0];
    for (uint32_t i = threadIdx.x; i < 16u * WL * 32; i += 256) dst[i] = src[i];
    SINK(sink);
}

// MODE 7: the write stream alone (no reads): its own cost when nothing competes
template <int WL>
__global__ __launch_bounds__(256) void k_pmc_write_only(uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf)
{
    uint32_t tb, g; map(G, tb, g);
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const uint64_t slot = (uint64_t)f * ntb * 4 + (uint64_t)tb * 4 + w;
        uint32_t *dst = wbuf + 64 + slot * 160;
        for (int l = 0; l < WL; ++l) if (lane < 32) dst[l * 32 + lane] = f + l;
    }
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const uint64_t bytes = 2ull << 30;
    u32x4 *p; uint32_t *out, *wbuf;
    hipMalloc(&p, bytes); hipMalloc(&out, 4); hipMemset(p, 1, bytes);
    const uint64_t frame16 = (32ull << 20) / 16; const uint32_t ntb = 1024, G = 16;
    hipMalloc(&wbuf, (uint64_t)64 * ntb * 4 * 640 + (64ull << 20));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto launch, const char *name) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %.3f ms\n", name, ms / reps);
    };
    const dim3 grid(ntb * G), blk(256);
    for (int round = 0; round < 2; ++round) {
        time([&] { hipLaunchKernelGGL(k_pmc_read, grid, blk, 0, 0, p, frame16, ntb, G, out); }, "read only");
        time([&] { hipLaunchKernelGGL((k_pmc_rw<1, 0>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 1 line, plain stores");
        time([&] { hipLaunchKernelGGL((k_pmc_rw<3, 0>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 3 lines, plain stores");
        time([&] { hipLaunchKernelGGL((k_pmc_rw<3, 1>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 3 lines, nontemporal stores");
        time([&] { hipLaunchKernelGGL((k_pmc_rw<3, 2>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 3 lines, sc0 sc1 stores");
        time([&] { hipLaunchKernelGGL((k_pmc_rw<3, 5>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 3 lines, 16-byte stores");
        time([&] { hipLaunchKernelGGL((k_pmc_rw<3, 6>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 3 lines, 16-byte sc0 sc1 stores");
        time([&] { hipLaunchKernelGGL((k_pmc_rw_onewriter<3>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 3 lines, one writer wave per WG");
        time([&] { hipLaunchKernelGGL((k_pmc_rw_wgburst<3>), grid, blk, 0, 0, p, frame16, ntb, G, wbuf); }, "read + 3 lines, one 6 KiB burst per WG");
        time([&] { hipLaunchKernelGGL((k_pmc_write_only<3>), grid, blk, 0, 0, ntb, G, wbuf); }, "3 lines, no reads");
    }
    hipDeviceSynchronize();
    return 0;
}
