// rc_hostdec.hip - the host half of the batched reader for streams only a STOCK decoder takes (DESIGN.md "Foreign streams"):
// files the reference's writer produced with zstandard / lz4.frame / zlib (pyrecode/recode_compressors.py:82-120).  Such a stream is
// one serial chain; the reference decodes it with one library call per stream (recode_compressors.py:40-79).  Here the 2 n streams
// of a batch go through the SAME libraries (libzstd / liblz4 / libz, bound at run time: they are not link-time dependencies of this
// library) on worker threads, each stream straight into its place of a stored-pieces image that rc_expand_frames (op_mode 0) then
// expands on the device.  No HIP call in this file.
#include <dlfcn.h>
#include <type_traits>

#include "rc_host.h"

namespace {

struct ZstdApi {
    void *(*createDCtx)() = nullptr;
    size_t (*freeDCtx)(void *) = nullptr;
    size_t (*decompressDCtx)(void *, void *, size_t, const void *, size_t) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    bool ok = false;
};
struct Lz4Api {
    size_t (*createCtx)(void **, unsigned) = nullptr;
    size_t (*freeCtx)(void *) = nullptr;
    size_t (*decompress)(void *, void *, size_t *, const void *, size_t *, const void *) = nullptr;
    void (*reset)(void *) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    bool ok = false;
};
struct ZlibApi {
    int (*uncompress)(unsigned char *, unsigned long *, const unsigned char *, unsigned long) = nullptr;
    bool ok = false;
};

template <class F>
bool bind(void *h, const char *name, F &fn)
{
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

const ZstdApi &zstd_api()
{
    static const ZstdApi api = [] {
        ZstdApi a;
        void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (h)
            a.ok = bind(h, "ZSTD_createDCtx", a.createDCtx) && bind(h, "ZSTD_freeDCtx", a.freeDCtx) &&
                   bind(h, "ZSTD_decompressDCtx", a.decompressDCtx) && bind(h, "ZSTD_isError", a.isError);
        return a;
    }();
    return api;
}
const Lz4Api &lz4_api()
{
    static const Lz4Api api = [] {
        Lz4Api a;
        void *h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
        if (h)
            a.ok = bind(h, "LZ4F_createDecompressionContext", a.createCtx) && bind(h, "LZ4F_freeDecompressionContext", a.freeCtx) &&
                   bind(h, "LZ4F_decompress", a.decompress) && bind(h, "LZ4F_resetDecompressionContext", a.reset) &&
                   bind(h, "LZ4F_isError", a.isError);
        return a;
    }();
    return api;
}
const ZlibApi &zlib_api()
{
    static const ZlibApi api = [] {
        ZlibApi a;
        void *h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
        if (h) a.ok = bind(h, "uncompress", a.uncompress);
        return a;
    }();
    return api;
}

// One decoding context per worker thread, kept for the thread's life (the workers never end): a context's buffers are 100 KB - 2 MB,
// i.e. mmap / munmap per stream if it were made per call.
struct ThreadCtx {
    void *zstd = nullptr, *lz4 = nullptr;
    ~ThreadCtx()
    {
        if (zstd) zstd_api().freeDCtx(zstd);
        if (lz4) lz4_api().freeCtx(lz4);
    }
};
thread_local ThreadCtx t_ctx;

// stream -> exactly `want` bytes at dst; false: the library rejected it or it has another length
bool decode_one(uint32_t scheme, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t want)
{
    if (scheme == 1) {
        const ZstdApi &z = zstd_api();
        if (!t_ctx.zstd && !(t_ctx.zstd = z.createDCtx())) return false;
        const size_t r = z.decompressDCtx(t_ctx.zstd, dst, want, src, n);   // (frames without a content-size field are fine: dst bounds it)
        return !z.isError(r) && r == want;
    }
    if (scheme == 2) {
        const Lz4Api &l = lz4_api();
        if (!t_ctx.lz4 && l.isError(l.createCtx(&t_ctx.lz4, 100))) { t_ctx.lz4 = nullptr; return false; }
        l.reset(t_ctx.lz4);
        uint64_t got = 0, pos = 0;
        bool ended = false;
        while (pos < n) {
            size_t dn = want - got, sn = n - pos;
            const size_t r = l.decompress(t_ctx.lz4, dst + got, &dn, src + pos, &sn, nullptr);
            if (l.isError(r)) break;
            got += dn;
            pos += sn;
            if (r == 0) { ended = true; break; }
            if (!dn && !sn) break;          // no progress: the frame wants more input or more room than there is
        }
        if (!ended) l.reset(t_ctx.lz4);     // (an abandoned frame leaves the context mid-stream)
        return ended && got == want && pos == n;
    }
    if (scheme == 0) {
        unsigned long dn = want;
        return zlib_api().uncompress(dst, &dn, src, n) == 0 && dn == want;
    }
    return false;
}

WorkerPool *g_dec_pool = new WorkerPool;     // its own pool: a decode that runs ahead must not queue behind the reader's header walk

}  // namespace

RC_EXPORT int rc_host_decoder_available(uint32_t scheme)
{
    return scheme == 1 ? zstd_api().ok : scheme == 2 ? lz4_api().ok : scheme == 0 ? zlib_api().ok : 0;
}

RC_EXPORT int rc_host_decode_streams(uint32_t scheme, const uint8_t *src, uint8_t *dst, const uint64_t *spans, uint32_t n, uint32_t threads)
{
    if (!rc_host_decoder_available(scheme)) return fail(RC_ERR_UNSUPPORTED, "rc_host_decode_streams: no stock decoder for this scheme on this host");
    if (!n) return RC_OK;
    if (!src || !dst || !spans) return fail(RC_ERR_BAD_ARG, "rc_host_decode_streams: null argument");
    // longest streams first, one at a time off a shared counter: the batch ends with short ones
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return spans[4 * a + 1] > spans[4 * b + 1]; });
    uint32_t hw = usable_cpus();   // (not hardware_concurrency(): a cgroup quota grants fewer)
    if (const char *e = getenv("RC_DECODE_THREADS")) hw = (uint32_t)atoi(e);
    uint32_t nthr = threads ? threads : std::min<uint32_t>(16, hw ? hw : 1);
    nthr = std::max<uint32_t>(1, std::min<uint32_t>(nthr, std::min<uint32_t>(n, 64)));
    std::atomic<uint32_t> next{0};
    std::atomic<int64_t> bad{-1};
    g_dec_pool->run(nthr, [&](uint32_t) {
        for (;;) {
            const uint32_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= n || bad.load(std::memory_order_relaxed) >= 0) return;
            const uint64_t *sp = spans + 4 * (uint64_t)order[k];
            if (!sp[3]) continue;                                          // nothing to regenerate (an empty value stream)
            if (!decode_one(scheme, src + sp[0], sp[1], dst + sp[2], sp[3])) bad.store(order[k], std::memory_order_relaxed);
        }
    });
    if (bad.load() >= 0) {
        char msg[160];
        snprintf(msg, sizeof msg, "rc_host_decode_streams: the stock decoder rejected stream %lld (or it does not decode to the expected size)",
                 (long long)bad.load());
        return fail(RC_ERR_CORRUPT, msg);
    }
    return RC_OK;
}

// (row, col, value) rows -> the three arrays of a COO matrix, one pass over the rows (three strided numpy conversions read them three times).
RC_EXPORT int rc_split_triplets(const uint64_t *triplets, uint64_t n, int32_t *row, int32_t *col, void *val, uint32_t val_bytes)
{
    if (n && (!triplets || !row || !col || !val)) return fail(RC_ERR_BAD_ARG, "rc_split_triplets: null argument");
    auto run = [&](auto *v) {
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t *t = triplets + 3 * i;
            row[i] = (int32_t)t[0];
            col[i] = (int32_t)t[1];
            v[i] = (typename std::remove_pointer<decltype(v)>::type)t[2];
        }
    };
    switch (val_bytes) {
    case 1: run(static_cast<uint8_t *>(val)); break;
    case 2: run(static_cast<uint16_t *>(val)); break;
    case 4: run(static_cast<uint32_t *>(val)); break;
    case 8: run(static_cast<uint64_t *>(val)); break;
    default: return fail(RC_ERR_BAD_ARG, "rc_split_triplets: val_bytes must be 1, 2, 4 or 8");
    }
    return RC_OK;
}
