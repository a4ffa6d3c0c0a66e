// Host-only build of the device block encoder (pyrecode_amd/csrc/rc_zstd_block.h) for CPU tests: lets the stock libzstd
// decoder judge the bitstream the HIP kernel will emit, without a GPU.  Test infrastructure, not a product path.
#include <cstring>
#include <vector>

#include "../../pyrecode_amd/csrc/rc_zstd_block.h"

extern "C" int64_t zstd_check_encode_frame(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, int streaming)
{
    static rc::ZstdTables T;
    static bool init = false;
    if (!init) { rc::zstd_build_tables(T); init = true; }
    uint8_t *p = dst;
    const uint8_t hdr[6] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x00};  // magic, FHD (no FCS, window descriptor follows), 1 KiB window
    if (cap < 6) return -1;
    memcpy(p, hdr, 6);
    p += 6;
    std::vector<rc::ZstdSeq> seq(rc::ZSTD_BLK / 4 + 2);
    uint8_t tmp[rc::ZSTD_SLOT_MAX + 16];
    if (n == 0) {  // a frame needs at least one block: empty raw block, last
        if ((uint64_t)(p - dst) + 3 > cap) return -1;
        p[0] = 1; p[1] = 0; p[2] = 0;
        return (p - dst) + 3;
    }
    alignas(16) uint32_t in32[rc::ZSTD_BLK / 4 + 8];
    alignas(16) uint8_t slot[rc::ZSTD_BLK + 16];
    for (uint64_t o = 0; o < n; o += rc::ZSTD_BLK) {
        const uint32_t len = (uint32_t)((n - o) < rc::ZSTD_BLK ? (n - o) : rc::ZSTD_BLK);
        uint32_t used;
        if (streaming) {  // the form the HIP kernel runs
            memset(in32, 0, sizeof in32);
            memcpy(in32, src + o, len);
            used = rc::zstd_encode_block_stream(in32, len, slot, rc::ZSTD_BLK + 16, T, o + len >= n);
            memcpy(tmp, slot, used);
        } else
            used = rc::zstd_encode_block(src + o, len, tmp, seq.data(), T, o + len >= n);
        if ((uint64_t)(p - dst) + used > cap) return -1;
        memcpy(p, tmp, used);
        p += used;
    }
    return p - dst;
}
