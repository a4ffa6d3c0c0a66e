"""Development: how the stock decoders scale over a thread pool on this host, into ordinary and into page-locked memory (what
ReCoDeReader._host_decode_batch does for streams a foreign encoder wrote).  usage: host_decode_scaling.py [scheme 1|2] [frames per batch]"""
import ctypes as C, ctypes.util, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from pyrecode_amd import _lib, recode_compressors as rcx

scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = np.random.default_rng(0)
N = 4096 * 4096
mask = rng.random(N) < 0.01
bitmap = np.packbits(mask, bitorder="little").tobytes()
packed = rng.integers(1, 2048, int(mask.sum()), dtype=np.uint16).tobytes()
if scheme == 1:
    z = C.CDLL(ctypes.util.find_library("zstd")); z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    def enc(b):
        dst = C.create_string_buffer(len(b) + len(b) // 8 + 1024); n = z.ZSTD_compress(dst, len(dst), b, len(b), 1); return dst.raw[:n]
else:
    lz = C.CDLL(ctypes.util.find_library("lz4"))
    lz.LZ4F_compressFrameBound.restype = C.c_size_t; lz.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
    lz.LZ4F_compressFrame.restype = C.c_size_t; lz.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    def enc(b):
        dst = C.create_string_buffer(lz.LZ4F_compressFrameBound(len(b), None) + 64); n = lz.LZ4F_compressFrame(dst, len(dst), b, len(b), None); return dst.raw[:n]
cb, cp = enc(bitmap), enc(packed)
dec = rcx.host_stream_decoder(scheme)
per = len(bitmap) + len(packed)
print("scheme %d: bitmap %d -> %d, values %d -> %d bytes; cpus %d (affinity %d)" % (scheme, len(cb), len(bitmap), len(cp), len(packed),
                                                                                  os.cpu_count(), len(os.sched_getaffinity(0))))
t = time.perf_counter()
o = np.empty(len(bitmap), np.uint8)
for _ in range(8): dec(memoryview(cb), o.size, o)
print("one thread, binary map alone: %.2f ms" % ((time.perf_counter() - t) / 8 * 1e3))
for kind in ("ordinary", "page-locked"):
    if kind == "ordinary":
        buf = np.empty(per * nfr, np.uint8)
    else:
        pin = _lib.PinnedBuffer(per * nfr + 64)
        buf = pin.array[:per * nfr]
    buf[:] = 0
    def one(i):
        f, which = divmod(i, 2)
        at = f * per
        if which: dec(memoryview(cp), len(packed), buf[at + len(bitmap):at + per])
        else: dec(memoryview(cb), len(bitmap), buf[at:at + len(bitmap)])
    for nt in (1, 2, 4, 8, 16, 32):
        pool = ThreadPoolExecutor(nt)
        list(pool.map(one, range(2 * nfr)))
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); list(pool.map(one, range(2 * nfr))); best = min(best, time.perf_counter() - t)
        pool.shutdown()
        print("  %-11s %2d threads: %6.0f frames/s" % (kind, nt, nfr / best))

# the library's own batch decoder (rc_host_decode_streams): no interpreter in the loop
L = _lib.lib()
src = np.frombuffer(cb + cp, np.uint8)
table = np.array([(0 if not w else len(cb), len(cp) if w else len(cb), f * per + (len(bitmap) if w else 0), len(packed) if w else len(bitmap))
                  for f in range(nfr) for w in (0, 1)], np.uint64)
for nt in (1, 2, 4, 8, 16, 32):
    best = 1e9
    for _ in range(4):
        t = time.perf_counter()
        st = L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(buf), _lib.ptr(table), table.shape[0], nt)
        best = min(best, time.perf_counter() - t)
    assert st == 0, _lib.last_error()
    print("  native      %2d threads: %6.0f frames/s" % (nt, nfr / best))
