"""Development helper: stage timings of the device-resident pipeline (not the bench contract).
usage: quick_perf.py ny nx B ppm depth scheme [level] [keep_maps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip
if os.environ.get("RC_AB_LIB"):  # A/B runs on one box: a second build of the library (development only)
    hip.LIB_PATH = os.path.abspath(os.environ["RC_AB_LIB"])

a = sys.argv[1:]
ny, nx, B, ppm, d, scheme = (int(v) for v in a[:6])
level = int(a[6]) if len(a) > 6 else 1
keep = int(a[7]) if len(a) > 7 else 0
N = nx * ny
L = hip.lib()
dark = torch.empty(N, dtype=torch.int16, device="cuda")
frames = torch.empty((B, N), dtype=torch.int16, device="cuda")
hip.check(L.rc_synth_dark(0, 1, N, dark.data_ptr()))
hip.check(L.rc_synth_frames(0, 1, 0, B, N, ppm, dark.data_ptr(), frames.data_ptr()))
ctx = hip.ReduceContext(nx, ny, d, level, 1, scheme, 1, 0, max_batch=B)
ctx.set_threshold(dark.data_ptr())
ctx.keep_binary_maps(bool(keep))
cap = B * (N // 2 + 4096)
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
recn = np.zeros(B + 1, np.uint64); mdn = np.zeros((B, 3), np.uint32)
runs = []
for it in range(24):
    hip.check(L.rc_reduce_compress_batch(ctx.handle, frames.data_ptr(), B, 0, out.data_ptr(), cap, recn.ctypes.data, mdn.ctypes.data))
    if it >= 4:
        runs.append(ctx.stage_ms())
runs = np.array(runs)
med, best = np.median(runs, axis=0), runs.min(axis=0)
print("shape %dx%d B=%d ppm=%d d=%d scheme=%d level=%d keep=%d: median ms [reduce, codec, scan, layout+assemble, total] = %s (min reduce %.3f total %.3f) -> %.0f frames/s, reduce %.2f TB/s" %
      (ny, nx, B, ppm, d, scheme, level, keep, ["%.3f" % v for v in med], best[0], best[4], B / med[4] * 1e3, B * N * 2 / med[0] / 1e9))
