"""Does the reduce kernel's time depend on WHERE its buffers were allocated?  One process, several allocations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip

ny = nx = 4096
B, N = 64, 4096 * 4096
L = hip.lib()
keep = []
for trial in range(8):
    dark = torch.empty(N, dtype=torch.int16, device="cuda")
    frames = torch.empty((B, N), dtype=torch.int16, device="cuda")
    hip.check(L.rc_synth_dark(0, 1, N, dark.data_ptr()))
    hip.check(L.rc_synth_frames(0, 1, 0, B, N, 10000, dark.data_ptr(), frames.data_ptr()))
    ctx = hip.ReduceContext(nx, ny, 16, 1, 1, 2, 1, 0, max_batch=B)
    ctx.set_threshold(dark.data_ptr())
    ctx.keep_binary_maps(False)
    cap = B * (N // 2 + 4096)
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    recn = np.zeros(B + 1, np.uint64); mdn = np.zeros((B, 3), np.uint32)
    runs = []
    for it in range(24):
        hip.check(L.rc_reduce_compress_batch(ctx.handle, frames.data_ptr(), B, 0, out.data_ptr(), cap, recn.ctypes.data, mdn.ctypes.data))
        if it >= 4:
            runs.append(ctx.stage_ms()[0])
    print("trial %d: frames @%x  reduce median %.3f min %.3f ms" % (trial, frames.data_ptr(), float(np.median(runs)), min(runs)), flush=True)
    if trial % 2 == 0:
        keep.append((dark, frames, ctx, out))   # hold on to some allocations so that the next ones land elsewhere
    else:
        ctx.close()
