"""Development: per-phase cycle shares of the reduce kernel (a -DRC_PHASE_TIMING build, tools/build_def.sh phase -DRC_PHASE_TIMING).
usage: RC_AB_LIB=ab_build/librecode_hip_phase.so python tools/phase_timing.py ny nx B ppm depth scheme [level] [clevel] [clustered]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip
hip.LIB_PATH = os.path.abspath(os.environ["RC_AB_LIB"])
a = sys.argv[1:]
ny, nx, B, ppm, d, scheme = (int(v) for v in a[:6])
level = int(a[6]) if len(a) > 6 else 1
clevel = int(a[7]) if len(a) > 7 else 1
clustered = len(a) > 8 and a[8] == 'clustered'
N = nx * ny
L = hip.lib()
dark = torch.empty(N, dtype=torch.int16, device="cuda")
frames = torch.empty((B, N), dtype=torch.int16, device="cuda")
hip.check(L.rc_synth_dark(0, 1, N, dark.data_ptr()))
if clustered:
    hip.check(L.rc_synth_frames_clustered(0, 1, 0, B, nx, ny, ppm, dark.data_ptr(), frames.data_ptr()))
else:
    hip.check(L.rc_synth_frames(0, 1, 0, B, N, ppm, dark.data_ptr(), frames.data_ptr()))
ctx = hip.ReduceContext(nx, ny, d, level, 1, scheme, clevel, 0, max_batch=B)
ctx.set_threshold(dark.data_ptr())
ctx.keep_binary_maps(False)
cap = int(L.rc_out_capacity(ctx.handle, B))
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
recn = np.zeros(B + 1, np.uint64); mdn = np.zeros((B, 3), np.uint32)
ph = (C.c_ulonglong * 16)()
for it in range(6):
    hip.check(L.rc_reduce_compress_batch(ctx.handle, frames.data_ptr(), B, 0, out.data_ptr(), cap, recn.ctypes.data, mdn.ctypes.data))
    if it == 1:
        L.rc_debug_phases(ph)   # clear after warm-up
L.rc_debug_phases(ph)
v = np.array(list(ph)[:7], np.float64)
ntf = max(int(ph[7]), 1)
names = ["wait for the frame's loads", "subtract + mask", "issue next loads", "stage values + transpose", "compaction (+ pack)", "block codec", "flush (stores)"]
print("shape %dx%d B=%d ppm=%d d=%d scheme=%d level=%d clevel=%d; reduce %.3f ms" % (ny, nx, B, ppm, d, scheme, level, clevel, ctx.stage_ms()[0]))
for n_, x in zip(names, v):
    print("  %-28s %6.1f %%   %7.0f memtime ticks per tile-frame" % (n_, 100 * x / v.sum(), x / ntf))
sub = np.array(list(ph)[8:12], np.float64)
if sub.sum() and scheme == 256:
    for n_, x in zip(["deflate: parse (rc_lz4_block.h's parsers)", "deflate: clear the image, build the units", "deflate: scan + LDS ORs", "deflate: tail (sync marker / stored)"], sub):
        print("    %-50s %7.0f ticks per tile-frame" % (n_, x / ntf))
elif sub.sum():
    for n_, x in zip(["LZ4: event parse (list, one / two events per lane)", "LZ4: run parse (blocks the event parser left)", "LZ4: sequence sizes + scan", "LZ4: emit (tokens, literals, offsets)"], sub):
        print("    %-50s %7.0f ticks per tile-frame" % (n_, x / ntf))
