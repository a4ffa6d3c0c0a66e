"""Development helper: stage timings of the device-resident pipeline at a given shape (not the bench contract)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip

ny, nx = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ppm = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
d = int(sys.argv[5]) if len(sys.argv) > 5 else 16
scheme = int(sys.argv[6]) if len(sys.argv) > 6 else 2
N = nx * ny
L = hip.lib()
dark = torch.empty(N, dtype=torch.int16, device="cuda")
frames = torch.empty((B, N), dtype=torch.int16, device="cuda")
hip.check(L.rc_synth_dark(0, 1, N, dark.data_ptr()))
hip.check(L.rc_synth_frames(0, 1, 0, B, N, ppm, dark.data_ptr(), frames.data_ptr()))
ctx = hip.ReduceContext(nx, ny, d, 1, 1, scheme, 1, 0, max_batch=B)
ctx.set_threshold(dark.data_ptr())
cap = B * (N // 4 + 4096)
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
rec = torch.empty(B + 1, dtype=torch.int64, device="cuda")
md = torch.empty(B * 3, dtype=torch.int32, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for it in range(3):
    ctx.enqueue(frames.data_ptr(), B, 0, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
ctx.sync()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 10
e0.record()
for it in range(K):
    ctx.enqueue(frames.data_ptr(), B, 0, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
e1.record()
ctx.sync()
ms = e0.elapsed_time(e1) / K
print("shape %dx%d B=%d ppm=%d d=%d scheme=%d: %.3f ms/batch  %.1f us/frame  %.0f frames/s  %.2f TB/s in" %
      (ny, nx, B, ppm, d, scheme, ms, ms * 1e3 / B, B / ms * 1e3, B * N * 2 / ms / 1e9))
print("record bytes/frame:", int(rec.cpu()[-1]) / B, "md[0]:", md.cpu()[:3].tolist())
# per-stage through the synchronous entry point with device pointers
recn = np.zeros(B + 1, np.uint64); mdn = np.zeros((B, 3), np.uint32)
hip.check(L.rc_reduce_compress_batch(ctx.handle, frames.data_ptr(), B, 0, out.data_ptr(), cap, recn.ctypes.data, mdn.ctypes.data))
print("stage ms [reduce, scan, bitmap-codec, layout+assemble, total]:", ["%.3f" % v for v in ctx.stage_ms()])
