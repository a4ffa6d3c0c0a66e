"""Development probe: where does host -> device ingest bandwidth go on this box?  (not part of the product)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from pyrecode_amd import _lib as hip
L = hip.lib()
rt = C.CDLL("libamdhip64.so")
GB = 1 << 30
n = 2 * GB
d = torch.empty(n, dtype=torch.uint8, device="cuda")
def h2d(ptr, label):
    torch.cuda.synchronize()
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        rt.hipMemcpy(C.c_void_p(d.data_ptr()), C.c_void_p(ptr), C.c_size_t(n), 1)
        torch.cuda.synchronize()
        best = max(best, n / (time.perf_counter() - t0) / 1e9)
    print("%-52s %.1f GB/s" % (label, best), flush=True)
a = np.ones(n, np.uint8)
h2d(a.ctypes.data, "hipMemcpy from pageable numpy")
pin = hip.PinnedBuffer(n); pin.array[:] = 1
h2d(pin.array.ctypes.data, "hipMemcpy from hipHostMalloc (default flags)")
hip.check(L.rc_host_register(a.ctypes.data, n))
h2d(a.ctypes.data, "hipMemcpy from hipHostRegister'ed numpy")
hip.check(L.rc_host_unregister(a.ctypes.data))
for flags, name in ((0x2000_0000, "NumaUser"), (0x4000_0000, "Coherent"), (0x8000_0000, "NonCoherent")):
    p = C.c_void_p()
    if rt.hipHostMalloc(C.byref(p), C.c_size_t(n), C.c_uint(flags)) == 0:
        C.memset(p, 1, n)
        h2d(p.value, "hipMemcpy from hipHostMalloc(%s)" % name)
        rt.hipHostFree(p)
    else:
        print("hipHostMalloc(%s) failed" % name)
src = np.ones(n, np.uint8)
for th in (1, 4, 8, 16, 32):
    pool = ThreadPoolExecutor(th)
    step = n // (th * 2)
    t0 = time.perf_counter()
    list(pool.map(lambda o: np.copyto(pin.array[o:o + step], src[o:o + step]), range(0, n, step)))
    print("host copy pageable -> pinned, %2d threads: %.1f GB/s" % (th, n / (time.perf_counter() - t0) / 1e9), flush=True)
    pool.shutdown()
print("cpus", len(os.sched_getaffinity(0)))
os.system("numactl -H 2>/dev/null | head -5; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3")
