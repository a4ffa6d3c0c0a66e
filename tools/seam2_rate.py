"""Development probe: seam 2 (rc_compress / rc_decompress on one host buffer per call) rates on a 4096x4096 1 % binary map."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrecode_amd import recode_compressors as rc
rng = np.random.default_rng(0)
bits = rng.random(4096 * 4096) < 0.01
bitmap = np.packbits(bits, bitorder="little")
for scheme, name in ((2, "lz4"), (1, "zstd"), (8, "blosc-lz4")):
    comp = rc.device_compress(scheme, 1, bitmap)
    t0 = time.perf_counter(); K = 20
    for _ in range(K): rc.device_compress(scheme, 1, bitmap)
    tc = (time.perf_counter() - t0) / K
    back = rc.device_decompress(scheme, comp, bitmap.size)
    assert back == bitmap.tobytes()
    t0 = time.perf_counter()
    for _ in range(K): rc.device_decompress(scheme, comp, bitmap.size)
    td = (time.perf_counter() - t0) / K
    print("%-10s %7d -> %7d B: compress %.2f ms (%.2f GB/s), decompress %.2f ms (%.2f GB/s)" % (name, bitmap.size, len(comp), tc * 1e3, bitmap.size / tc / 1e9, td * 1e3, bitmap.size / td / 1e9))
