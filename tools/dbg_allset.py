import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from pyrecode_amd import _lib as hip
for ny, nx in ((512, 512), (4096, 4096)):
    N, B = ny * nx, 3
    dark = torch.empty(N, dtype=torch.int16, device="cuda"); frames = torch.empty((B, N), dtype=torch.int16, device="cuda")
    hip.check(hip.lib().rc_synth_dark(0, 5, N, dark.data_ptr())); hip.check(hip.lib().rc_synth_frames(0, 5, 0, B, N, 10000, dark.data_ptr(), frames.data_ptr()))
    frames[1] = 30000
    torch.cuda.synchronize()
    for scheme in (2, 1, 0):
        ctx = hip.ReduceContext(nx, ny, 16, 1, 1 if scheme else 0, scheme, 1, 0, max_batch=B)
        ctx.set_dark(dark.data_ptr(), 0)
        cap = int(hip.lib().rc_out_capacity(ctx.handle, B))
        out = torch.empty(cap, dtype=torch.uint8, device="cuda"); rec = torch.zeros(B + 1, dtype=torch.int64, device="cuda"); md = torch.zeros((B, 3), dtype=torch.int32, device="cuda")
        ctx.enqueue(frames.data_ptr(), B, 0, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
        try:
            ctx.sync(); print(ny, scheme, "NO RAISE", rec.cpu().numpy(), md.cpu().numpy().view(np.uint32).tolist(), "frame bytes", N * 2)
        except Exception as e:
            print(ny, scheme, "raised", type(e).__name__, e)
        ctx.close()
