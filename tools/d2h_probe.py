"""Development probe: device -> host copy rates into page-locked memory (not part of the product)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrecode_amd import _lib as hip
L = hip.lib()
rt = C.CDLL("libamdhip64.so")
for n in (64 << 20, 1 << 30):
    d = torch.ones(n, dtype=torch.uint8, device="cuda")
    def d2h(ptr, label):
        torch.cuda.synchronize()
        best = 0
        for _ in range(4):
            t0 = time.perf_counter()
            rt.hipMemcpy(C.c_void_p(ptr), C.c_void_p(d.data_ptr()), C.c_size_t(n), 2)
            torch.cuda.synchronize()
            best = max(best, n / (time.perf_counter() - t0) / 1e9)
        print("%4d MB %-52s %.1f GB/s" % (n >> 20, label, best), flush=True)
    pin = hip.PinnedBuffer(n); pin.array[:] = 1
    d2h(pin.array.ctypes.data, "hipMemcpy to hipHostMalloc (default flags)")
    for flags, name in ((0x2000_0000, "NumaUser"), (0x4000_0000, "Coherent"), (0x8000_0000, "NonCoherent"), (0x1, "Portable"), (0x4, "WriteCombined")):
        p = C.c_void_p()
        if rt.hipHostMalloc(C.byref(p), C.c_size_t(n), C.c_uint(flags)) == 0:
            C.memset(p, 1, n)
            d2h(p.value, "hipMemcpy to hipHostMalloc(%s)" % name)
            rt.hipHostFree(p)
        else:
            print("hipHostMalloc(%s) failed" % name)
    t = torch.empty(n, dtype=torch.uint8).pin_memory()
    d2h(t.data_ptr(), "hipMemcpy to torch pinned")
    del d
