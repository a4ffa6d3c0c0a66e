// Development probe: HBM read bandwidth for the access shapes the reduce kernel uses (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *out)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 * 8 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint64_t j = i + (uint64_t)r * 256;
        if (j < n16) { u32x4 v = NT ? __builtin_nontemporal_load(p + j) : p[j]; acc += v; }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
// frame-strided: workgroup b reads tile (b / G) of frames (b % G)*4 .. +4, like k_reduce_tiles' mapping
__global__ __launch_bounds__(256) void k_read_strided(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *out, int mode)
{
    uint32_t tb, g;
    if (mode == 0) { const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3; g = j % G; tb = (j / G) * 8 + xcd; }
    else { tb = blockIdx.x % ntb; g = blockIdx.x / ntb; }
    if (tb >= ntb) return;
    u32x4 acc = {0, 0, 0, 0};
    for (int z = 0; z < 4; ++z) {
        const u32x4 *fr = p + (uint64_t)(g * 4 + z) * frame16 + (uint64_t)tb * 2048 + (threadIdx.x >> 6) * 512 + (threadIdx.x & 63);
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
// like the reduce kernel's traffic: per (tile, frame) a wave reads 8 KiB and writes `wlines` full 128-byte lines into a slot
// (slot stride 8 KiB, like pix_slots) -> how much does a small write stream cost the read stream?
__global__ __launch_bounds__(256) void k_read_write(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf,
                                                    int wlines, int scalar_too)
{
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3; const uint32_t g = j % G; const uint32_t tb = (j / G) * 8 + xcd;
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const u32x4 *fr = p + (uint64_t)f * frame16 + (uint64_t)tb * 2048 + w * 512 + lane;
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
        const uint64_t slot = ((uint64_t)f * ntb * 4 + tb * 4 + w);  // tile-frame index
        uint32_t *dst = wbuf + slot * 2048;                              // 8 KiB slots
        for (int l = 0; l < wlines; ++l) if (lane < 32) dst[l * 32 + lane] = acc[0] + l;  // one full line per iteration
        if (scalar_too && lane == 0) wbuf[(uint64_t)64 * ntb * 4 * 2048 + slot] = acc[1];  // 4-byte count store
    }
}
// same, but the slot index order and stride are parameters: order 0 = [frame][tile] (frame-major), 1 = [tile][frame]
__global__ __launch_bounds__(256) void k_read_write2(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf,
                                                     int wlines, uint32_t stride_dw, int order, int nframes)
{
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3; const uint32_t g = j % G; const uint32_t tb = (j / G) * 8 + xcd;
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const u32x4 *fr = p + (uint64_t)f * frame16 + (uint64_t)tb * 2048 + w * 512 + lane;
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
        const uint64_t tile = (uint64_t)tb * 4 + w;
        const uint64_t slot = order == 0 ? (uint64_t)f * ntb * 4 + tile : tile * nframes + f;
        uint32_t *dst = wbuf + slot * stride_dw;
        for (int l = 0; l < wlines; ++l) if (lane < 32) dst[l * 32 + lane] = acc[0] + l;
    }
}
// same as k_read_write2 with nontemporal stores
__global__ __launch_bounds__(256) void k_read_write_nt(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf,
                                                       int wlines, uint32_t stride_dw, int nframes, int wide)
{
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3; const uint32_t g = j % G; const uint32_t tb = (j / G) * 8 + xcd;
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const u32x4 *fr = p + (uint64_t)f * frame16 + (uint64_t)tb * 2048 + w * 512 + lane;
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
        const uint64_t tile = (uint64_t)tb * 4 + w;
        const uint64_t slot = (uint64_t)f * ntb * 4 + tile;
        uint32_t *dst = wbuf + slot * stride_dw;
        if (wide) {  // 16-byte stores: 8 lanes per line
            if (lane < 8u * wlines) __builtin_nontemporal_store(acc, reinterpret_cast<u32x4 *>(dst) + lane);
        } else {
            for (int l = 0; l < wlines; ++l) if (lane < 32) __builtin_nontemporal_store(acc[0] + l, dst + l * 32 + lane);
        }
    }
}
// writes deferred to the end of the wave's 4 frames: one burst of 4 * wlines lines instead of 4 bursts of wlines
__global__ __launch_bounds__(256) void k_read_write_deferred(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf,
                                                             int wlines, uint32_t stride_dw, int contiguous)
{
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3; const uint32_t g = j % G; const uint32_t tb = (j / G) * 8 + xcd;
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t keep[4];
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const u32x4 *fr = p + (uint64_t)f * frame16 + (uint64_t)tb * 2048 + w * 512 + lane;
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
        keep[z] = acc[0] + acc[1];
    }
    const uint64_t tile = (uint64_t)tb * 4 + w;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        // contiguous: the 4 frames' outputs of this (tile, group) sit next to each other; else per-frame slots as before
        const uint64_t slot = contiguous ? (tile * G + g) * 4 + z : (uint64_t)f * ntb * 4 + tile;
        uint32_t *dst = wbuf + slot * stride_dw;
        for (int l = 0; l < wlines; ++l) if (lane < 32) dst[l * 32 + lane] = keep[z] + l;
    }
}
// the reduce kernel's own write pattern: per (tile, frame) one line into an 8 KiB-strided slot array, two lines into a
// 640 B-strided one, and two 4-byte stores into dense arrays - against the same bytes as three contiguous lines of ONE slot
__global__ __launch_bounds__(256) void k_read_write_pattern(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf,
                                                            int merged)
{
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3; const uint32_t g = j % G; const uint32_t tb = (j / G) * 8 + xcd;
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t nslots = (uint64_t)64 * ntb * 4;
    uint32_t *A = wbuf, *Bs = wbuf + nslots * 2048, *C = Bs + nslots * 160, *D = C + nslots;
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const u32x4 *fr = p + (uint64_t)f * frame16 + (uint64_t)tb * 2048 + w * 512 + lane;
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
        const uint64_t slot = (uint64_t)f * ntb * 4 + (uint64_t)tb * 4 + w;
        if (merged == 1) {
            uint32_t *dst = A + slot * 2208;   // 8832-byte slots
            for (int l = 0; l < 3; ++l) if (lane < 32) dst[l * 32 + lane] = acc[0] + l;
        } else {
            if (lane < 32) A[slot * 2048 + lane] = acc[0];
            Bs[slot * 160 + lane] = acc[1];
            if (lane == 0 && merged != 2) { C[slot] = acc[2]; D[slot] = acc[3]; }
        }
    }
}
// outputs appended to one log per XCD (atomic cursor): what is written at about the same time lies next to each other
__global__ __launch_bounds__(256) void k_read_write_log(const u32x4 *__restrict__ p, uint64_t frame16, uint32_t ntb, uint32_t G, uint32_t *__restrict__ wbuf,
                                                        unsigned long long *__restrict__ cursor, uint64_t log_dw, int wlines, int nlogs)
{
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3; const uint32_t g = j % G; const uint32_t tb = (j / G) * 8 + xcd;
    if (tb >= ntb) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t lg = nlogs == 8 ? xcd : (nlogs == 1 ? 0u : (blockIdx.x % (uint32_t)nlogs));
    for (int z = 0; z < 4; ++z) {
        const uint32_t f = g * 4 + z;
        const u32x4 *fr = p + (uint64_t)f * frame16 + (uint64_t)tb * 2048 + w * 512 + lane;
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += __builtin_nontemporal_load(fr + r * 64);
        unsigned long long pos = 0;
        if (lane == 0) pos = atomicAdd(&cursor[lg * 16], (unsigned long long)(wlines * 32));   // dwords
        pos = __shfl(pos, 0);
        uint32_t *dst = wbuf + (uint64_t)lg * log_dw + pos;
        for (int l = 0; l < wlines; ++l) if (lane < 32) dst[l * 32 + lane] = acc[0] + l;
    }
}
int main()
{
    const uint64_t bytes = 2ull << 30, n16 = bytes / 16;
    u32x4 *p; uint32_t *out;
    hipMalloc(&p, bytes); hipMalloc(&out, 4); hipMemset(p, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto launch, const char *name) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-40s %.3f ms  %.2f TB/s\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
    };
    const uint32_t grid = (uint32_t)(n16 / (256 * 8));
    time([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, p, n16, out); }, "linear read, nontemporal");
    time([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, p, n16, out); }, "linear read, default policy");
    const uint64_t frame16 = (32ull << 20) / 16; const uint32_t ntb = 1024, G = 16;
    time([&] { hipLaunchKernelGGL(k_read_strided, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, out, 0); }, "frame-strided (groups adjacent, xcd)");
    time([&] { hipLaunchKernelGGL(k_read_strided, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, out, 1); }, "frame-strided (tiles adjacent)");
    uint32_t *wbuf; hipMalloc(&wbuf, (uint64_t)64 * ntb * 4 * (8192 + 1280) + (64ull * ntb * 4 * 4) + (64ull << 20));  // slack: offset / stride sweeps
    for (int wl = 0; wl <= 4; ++wl) for (int sc = 0; sc <= 1; ++sc) {
        char name[64]; snprintf(name, sizeof name, "read + %d full lines/tile%s", wl, sc ? " + 4B scalar" : "");
        time([&] { hipLaunchKernelGGL(k_read_write, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, wl, sc); }, name);
    }
    for (int order = 0; order <= 1; ++order) for (uint32_t stride : {96u, 128u, 160u, 2048u}) for (int wl : {1, 3}) {
        if (wl * 32 > (int)stride) continue;
        char name[80]; snprintf(name, sizeof name, "read + %d lines, slot stride %u B, %s", wl, stride * 4, order ? "[tile][frame]" : "[frame][tile]");
        time([&] { hipLaunchKernelGGL(k_read_write2, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, wl, stride, order, 64); }, name);
    }
    if (getenv("BW_LOG")) {
        unsigned long long *cursor; hipMalloc(&cursor, 1024 * 16 * 8);
        const uint64_t log_dw = (uint64_t)64 * ntb * 4 * 3 * 32;   // room for every tile-frame's 3 lines in ONE log (dwords)
        for (int rep = 0; rep < 2; ++rep) for (int nlogs : {8, 64, 256, 1024}) for (int wl : {1, 3}) {
            const uint64_t ldw = nlogs <= 8 ? log_dw : (log_dw / nlogs) * 2;   // per-log capacity (dwords), 2x the even share
            char name[80]; snprintf(name, sizeof name, "read + %d lines appended to %d log(s)", wl, nlogs);
            time([&] { hipMemsetAsync(cursor, 0, 1024 * 16 * 8, 0); hipLaunchKernelGGL(k_read_write_log, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, cursor, ldw, wl, nlogs); }, name);
        }
        for (int wl : {1, 3}) {
            char name[80]; snprintf(name, sizeof name, "read + %d lines, slot stride 640 B (reference)", wl);
            time([&] { hipLaunchKernelGGL(k_read_write2, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, wl, 160u, 0, 64); }, name);
        }
        return 0;
    }
    if (getenv("BW_PATTERN")) {
        for (int rep = 0; rep < 3; ++rep) for (int merged = 0; merged <= 2; ++merged) {
            char name[80]; snprintf(name, sizeof name, "reduce-kernel write pattern, %s", merged == 1 ? "merged into one slot (3 lines)" : merged == 2 ? "without the two 4-byte stores" : "as it is (4 streams)");
            time([&] { hipLaunchKernelGGL(k_read_write_pattern, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, merged); }, name);
        }
        return 0;
    }
    if (getenv("BW_OFFSET_SWEEP")) {
        // does the cost of the write stream depend on where it lies relative to the read stream?  (8 KiB slot stride = the
        // read stream's tile stride, so the relation is the same for every tile)
        printf("p = %p  wbuf = %p\n", (void *)p, (void *)wbuf);
        for (uint32_t off : {0u, 128u, 256u, 512u, 1024u, 2048u, 4096u, 4096u + 128u, 8192u, 16384u, 32768u, 65536u, 131072u, 262144u, 524288u,
                             1048576u, 2097152u, 4194304u, 8388608u, 16777216u}) {
            char name[80]; snprintf(name, sizeof name, "read + 1 line, stride 8 KiB, write base + %u", off);
            uint32_t *wb = wbuf + off / 4;
            time([&] { hipLaunchKernelGGL(k_read_write2, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wb, 1, 2048u, 0, 64); }, name);
        }
        for (uint32_t stride : {2048u, 2048u + 32u, 2048u + 64u, 2048u + 96u, 2048u + 160u, 2048u + 288u}) {
            char name[80]; snprintf(name, sizeof name, "read + 1 line, slot stride %u B", stride * 4);
            time([&] { hipLaunchKernelGGL(k_read_write2, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, 1, stride, 0, 64); }, name);
        }
        return 0;
    }
    for (int contiguous = 0; contiguous <= 1; ++contiguous) for (int wl : {1, 3}) {
        char name[80]; snprintf(name, sizeof name, "read 4 frames, then %d x 4 lines%s", wl, contiguous ? " contiguous" : "");
        time([&] { hipLaunchKernelGGL(k_read_write_deferred, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, wl, 160u, contiguous); }, name);
    }
    for (int wide = 0; wide <= 1; ++wide) for (int wl : {1, 3}) {
        char name[80]; snprintf(name, sizeof name, "read + %d lines NT stores%s, stride 640 B", wl, wide ? " (16-B)" : "");
        time([&] { hipLaunchKernelGGL(k_read_write_nt, dim3(ntb * G), dim3(256), 0, 0, p, frame16, ntb, G, wbuf, wl, 160u, 64, wide); }, name);
    }
    return 0;
}
