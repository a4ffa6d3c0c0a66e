// rc_blosc.hip - blosc1 chunk (bit-shuffle + LZ4, compression_scheme 8) block encoding on the GPU.
//
// Replaces `blosc.compress(data, clevel, cname='lz4', shuffle=blosc.BITSHUFFLE)` on the packed binary map
// (pyrecode/recode_compressors.py:108, called from recode_writer.py:503-505).  python-blosc is not a pinned dependency of
// the reference and is absent from the build image: the contract is a well-formed blosc1 chunk
// (c-blosc 1.x README_CHUNK_FORMAT / blosc.h) whose blocks are LZ4 blocks of the bit-shuffled data:
//   header 16 B: version 2 | versionlz 1 | flags | typesize 8 | nbytes | blocksize | cbytes   (little-endian int32s)
//   flags = 0x04 bit-shuffle | 0x10 blocks not split | 0x20 LZ4 format;   then int32 bstarts[nblocks];   then per block
//   int32 csize + csize bytes (csize == block bytes means "stored").
// One 512-byte tile = one block = 64 elements of typesize 8 (python-blosc's default typesize, the reference passes none).
//
// Bit-shuffle of a block (bitshuffle's bshuf_trans_bit_elem, little-endian bit order): with S elements, output row r
// (r = 0..63, bit r%8 of byte r/8 of every element) is S/8 bytes whose bit i is that bit of element i.  A lane owns one
// element (8 consecutive bytes), so row r IS the wave ballot of bit r.  S is rounded down to a multiple of 8; the bytes
// behind the shuffled part are copied unchanged (c-blosc's blosc_internal_bitshuffle).
#include "rc_launch.h"
#include "rc_lz4_block.h"

namespace rc {


// grid (ceil(ntiles/WAVES), B): wave w encodes block t = blockIdx.x*WAVES + w of frame blockIdx.y from the raw bitmap row.
__global__ __launch_bounds__(WG) void k_blosc_blocks(Scratch sc)
{
    __shared__ Lz4Lds s_lz[WAVES];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const uint32_t t = blockIdx.x * WAVES + w;
    const uint32_t f = blockIdx.y;
    if (t >= sc.ntiles) return;
    const uint64_t b0 = (uint64_t)t * TILE_BM;
    const uint32_t n = (uint32_t)min((uint64_t)TILE_BM, sc.nb - b0);
    const u32x2 v = reinterpret_cast<const u32x2 *>(sc.bitmap + (uint64_t)f * sc.nb_stride + b0)[lane];  // rows are padded
    const uint64_t elem = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    Lz4Lds &L = s_lz[w];
    reinterpret_cast<u32x2 *>(L.raw)[lane] = v;
    const uint64_t own = bitshuffle_block(elem, n, L);
    const uint32_t csize = lz4_encode_block(own, n, L);
    const uint64_t ft = (uint64_t)f * sc.ntiles + t;
    uint8_t *slot = sc.blk_slots + ft * sc.blk_stride;
    const uint32_t used = lz4_store_block(slot, own, n, csize, L, true);   // (blosc marks a stored block by csize == size)
    if (lane == 0) sc.blk_size[ft] = used;
}

void launch_blosc_encode_blocks(const Scratch &sc, uint32_t B, hipStream_t s)
{
    hipLaunchKernelGGL(k_blosc_blocks, dim3((sc.ntiles + WAVES - 1) / WAVES, B), dim3(WG), 0, s, sc);
}

// Stand-alone chunk of an arbitrary buffer (seam 2): header + bstarts + blocks.  One wavefront per block copy.
__global__ __launch_bounds__(WG) void k_blosc_gather(Scratch sc, uint8_t *__restrict__ out)
{
    const uint32_t t = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (t >= sc.ntiles) return;
    const uint32_t tab = 16 + 4 * sc.ntiles;
    if (t == 0 && lane_id() == 0) {
        const uint32_t nbytes = (uint32_t)sc.nb, bs = nbytes < (uint32_t)TILE_BM ? nbytes : (uint32_t)TILE_BM, cb = tab + sc.frame_cbytes[0];
        out[0] = 2; out[1] = 1; out[2] = 0x34; out[3] = 8;
        for (int k = 0; k < 4; ++k) { out[4 + k] = (uint8_t)(nbytes >> (8 * k)); out[8 + k] = (uint8_t)(bs >> (8 * k)); out[12 + k] = (uint8_t)(cb >> (8 * k)); }
    }
    const uint32_t off = tab + sc.blk_off[t];
    if (lane_id() == 0) for (int k = 0; k < 4; ++k) out[16 + 4 * t + k] = (uint8_t)(off >> (8 * k));
    const uint8_t *src = sc.blk_slots + (uint64_t)t * BLK_SLOT;
    const uint32_t n = sc.blk_size[t];
    for (uint32_t i = lane_id(); i < n; i += 64) out[off + i] = src[i];
}
void launch_blosc_gather(const Scratch &sc, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_blosc_gather, dim3((sc.ntiles + WAVES - 1) / WAVES), dim3(WG), 0, s, sc, out);
}

// ---- decode side (seam 2: de_compress, recode_compressors.py:61-76): the blocks' LZ4 streams are decoded by k_lz4_decode
// (rc_lz4.hip) into a scratch image of the shuffled chunk; this kernel undoes the shuffle.  One thread per output byte.
// shuffle: 0 none, 1 byte shuffle, 4 bit shuffle (the header's flag bits).
__global__ void k_blosc_unshuffle(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t nbytes, uint32_t blocksize,
                                  uint32_t typesize, uint32_t shuffle)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nbytes) return;
    const uint64_t b0 = (g / blocksize) * blocksize;
    const uint32_t bsize = (uint32_t)min((uint64_t)blocksize, nbytes - b0);
    const uint32_t p = (uint32_t)(g - b0);
    const uint8_t *blk = in + b0;
    uint32_t v = blk[p];
    if (shuffle == 4 && bsize >= typesize) {
        const uint32_t S = (bsize / typesize) & ~7u;
        if (p < S * typesize) {
            const uint32_t i = p / typesize, k = p % typesize, rowb = S >> 3;
            v = 0;
            for (uint32_t b = 0; b < 8; ++b) v |= ((blk[(8 * k + b) * rowb + (i >> 3)] >> (i & 7)) & 1u) << b;
        }
    } else if (shuffle == 1 && typesize > 1) {
        const uint32_t ne = bsize / typesize;
        if (p < ne * typesize) v = blk[(p % typesize) * ne + p / typesize];
    }
    out[g] = (uint8_t)v;
}
void launch_blosc_unshuffle(const uint8_t *in, uint8_t *out, uint64_t nbytes, uint32_t blocksize, uint32_t typesize,
                            uint32_t shuffle, hipStream_t s)
{
    if (!nbytes) return;
    hipLaunchKernelGGL(k_blosc_unshuffle, dim3((uint32_t)((nbytes + 255) / 256)), dim3(256), 0, s, in, out, nbytes, blocksize,
                       typesize, shuffle);
}

}  // namespace rc
