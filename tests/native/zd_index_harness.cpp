// Host-only harness (test infrastructure) around the batched reader's frame walker, pyrecode_amd/csrc/rc_zstd_dec.h::zd_index_frame:
// reads cases "[u32 n][u32 expect_regen][u64 total_expected][n bytes]" from a file, walks each one from an exact-size heap copy (so
// that AddressSanitizer sees any read past the stream) and prints "status blocks regen" per case.  Built with -fsanitize=address,undefined
// by tests/test_zstd_index_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../pyrecode_amd/csrc/rc_zstd_dec.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    for (;;) {
        uint32_t n, expect;
        uint64_t total;
        if (fread(&n, 4, 1, f) != 1) break;
        if (fread(&expect, 4, 1, f) != 1 || fread(&total, 8, 1, f) != 1) return 3;
        uint8_t *buf = (uint8_t *)malloc(n ? n : 1);
        if (n && fread(buf, 1, n, f) != n) return 3;
        std::vector<rc::ZdBlock> blocks;
        rc::ZdTables *T = new rc::ZdTables;
        uint64_t regen = 0;
        const int st = rc::zd_index_frame(buf, 0, n, 0, expect, total, blocks, *T, &regen);
        uint64_t sum = 0;
        for (const rc::ZdBlock &b : blocks) {
            if (st == rc::ZD_OK && (b.src + b.csize > n || b.dst != sum)) { printf("BAD entry\n"); return 4; }
            sum += b.regen;
        }
        printf("%d %zu %llu\n", st, blocks.size(), (unsigned long long)(st == rc::ZD_OK ? regen : 0));
        delete T;
        free(buf);
    }
    fclose(f);
    return 0;
}
