// rc_zstd.hip - Zstandard block encoding of packed binary maps on the GPU: one LANE per 512-byte block (the FSE bitstream
// of a block is a serial chain, so blocks - not bytes - are the unit of parallelism: 64 blocks per wavefront, 4096 blocks
// per 4096x4096 frame).  Encoder logic and format notes: rc_zstd_block.h (shared with the host-side format check).
//
// Replaces `ZstdCompressor(level, write_content_size=False).compress(bitmap)` (pyrecode/recode_writer.py:175-178,
// recode_compressors.py:88).  Decoding stays with the stock library on the host (recode_compressors.py:46).
#include <cstring>

#include "rc_launch.h"
#include "rc_zstd_block.h"

namespace rc {

// grid (ceil(ntiles/WG), B): thread t encodes block t of frame blockIdx.y from the raw bitmap row into blk_slots / blk_size.
__global__ __launch_bounds__(WG) void k_zstd_blocks(Scratch sc, const ZstdTables *__restrict__ tables)
{
    __shared__ ZstdTables T;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&T);
        for (uint32_t i = threadIdx.x; i < sizeof(ZstdTables) / 4; i += WG) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t t = blockIdx.x * WG + threadIdx.x;
    const uint32_t f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint64_t b0 = (uint64_t)t * TILE_BM;
    const uint32_t n = (uint32_t)min((uint64_t)TILE_BM, sc.nb - b0);
    const uint32_t *src32 = reinterpret_cast<const uint32_t *>(sc.bitmap + (uint64_t)f * sc.nb_stride + b0);
    const uint64_t ft = (uint64_t)f * sc.ntiles + t;
    sc.blk_size[ft] = zstd_encode_block_stream(src32, n, sc.blk_slots + ft * BLK_SLOT, BLK_SLOT, T, t + 1 == sc.ntiles);
}

void launch_zstd_encode_blocks(const Scratch &sc, uint32_t B, const void *tables_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_zstd_blocks, dim3((sc.ntiles + WG - 1) / WG, B), dim3(WG), 0, s, sc,
                       reinterpret_cast<const ZstdTables *>(tables_dev));
}

size_t zstd_tables_bytes() { return sizeof(ZstdTables); }
void zstd_tables_host(void *dst)
{
    ZstdTables t;
    zstd_build_tables(t);
    memcpy(dst, &t, sizeof t);
}

// Stand-alone frame of an arbitrary buffer (seam 2): header + the encoded blocks, nothing behind them (the final block
// carries Last_Block).  One wavefront per block copy.
__global__ __launch_bounds__(WG) void k_zstd_gather(Scratch sc, uint8_t *__restrict__ out)
{
    const uint32_t t = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (t >= sc.ntiles) return;
    if (t == 0 && lane_id() == 0) {
        out[0] = 0x28; out[1] = 0xB5; out[2] = 0x2F; out[3] = 0xFD;
        out[4] = 0x00;  // no content size, window descriptor follows
        out[5] = 0x00;  // 1 KiB window (blocks are 512 bytes, offsets are 1)
    }
    const uint8_t *src = sc.blk_slots + (uint64_t)t * BLK_SLOT;
    uint8_t *dst = out + 6 + sc.blk_off[t];
    const uint32_t n = sc.blk_size[t];
    for (uint32_t i = lane_id(); i < n; i += 64) dst[i] = src[i];
}
void launch_zstd_gather(const Scratch &sc, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_zstd_gather, dim3((sc.ntiles + WAVES - 1) / WAVES), dim3(WG), 0, s, sc, out);
}

}  // namespace rc
