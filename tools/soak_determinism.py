"""Development: the reduce -> records path run again and again over the same device-resident batch; every run's records must be
byte-identical to the first (the pipeline has no dispatch-order dependence by design, and a register / LDS hazard would show here).
usage: soak_determinism.py [seconds per configuration]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
L = hip.lib()
CFGS = [  # ny, nx, B, ppm, depth, scheme, level, clevel
    (4096, 4096, 64, 10000, 16, 2, 1, 1), (4096, 4096, 64, 10000, 16, 1, 1, 1), (4096, 4096, 64, 10000, 12, 2, 1, 0),
    (4096, 4096, 32, 100000, 12, 2, 1, 1), (4096, 4096, 32, 1000, 16, 8, 2, 1), (8184, 11520, 8, 50000, 12, 1, 1, 1),
    (4096, 4096, 64, 10000, 16, 2, 3, 1), (1000, 1003, 16, 30000, 10, 1, 1, 1),
    (3710, 3838, 32, 10000, 12, 2, 1, 1), (1023, 1023, 128, 20000, 12, 2, 1, 1), (4096, 4096, 32, 20000, 12, 2, 1, 1),   # N % 8 = 4, odd N, the two-events-per-lane parser
    (4096, 4096, 32, 10000, 12, 2, 2, 1), (2000, 3000, 16, 60000, 16, 1, 2, 1), (512, 512, 9, 145000, 12, 0, 1, 1)]        # round 5: level 2 at 1 % / dense (unions race for roots; results must not), small items
for ny, nx, B, ppm, d, scheme, level, clevel in CFGS:
    N = ny * nx
    dark = torch.empty(N, dtype=torch.int16, device="cuda")
    frames = torch.empty((B, N), dtype=torch.int16, device="cuda")
    hip.check(L.rc_synth_dark(0, 9, N, dark.data_ptr()))
    hip.check(L.rc_synth_frames(0, 9, 0, B, N, ppm, dark.data_ptr(), frames.data_ptr()))
    if N % 8:   # an unaligned view exercises the plain-load instantiation
        pass
    ctx = hip.ReduceContext(nx, ny, d, level, 1, scheme, clevel, 0, max_batch=B)
    ctx.set_dark(dark.data_ptr(), 0)
    ctx.keep_binary_maps(False)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_pipelined(True)
    cap = B * N
    outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
    recs = [torch.zeros(B + 1, dtype=torch.int64, device="cuda") for _ in range(2)]
    mds = [torch.zeros((B, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
    w = torch.arange(1, 4097, dtype=torch.int64, device="cuda")

    def digest(k):
        n = int(recs[k][-1].item())
        v = outs[k][:n].to(torch.int64)
        pad = (-n) % 4096
        if pad:
            v = torch.cat([v, torch.zeros(pad, dtype=torch.int64, device="cuda")])
        return n, int(v.sum().item()), int((v.view(-1, 4096) * w).sum().item()), mds[k].cpu().numpy().tobytes()
    with torch.cuda.stream(stream):
        for k in range(2):   # (zstd: the first batch fits the model)
            ctx.enqueue(frames.data_ptr(), B, 0, outs[k].data_ptr(), cap, recs[k].data_ptr(), mds[k].data_ptr())
        ctx.sync()
        want = digest(1)
        t0, runs, bad = time.time(), 0, 0
        while time.time() - t0 < budget:
            for k in range(2):
                ctx.enqueue(frames.data_ptr(), B, 0, outs[k].data_ptr(), cap, recs[k].data_ptr(), mds[k].data_ptr())
            ctx.sync()
            for k in range(2):
                runs += 1
                if digest(k) != want:
                    bad += 1
    ctx.close()
    print("%dx%d B=%d ppm=%d d=%d scheme=%d level=%d clevel=%d: %d runs, %d differ from the first, %d record bytes" % (ny, nx, B, ppm, d, scheme, level, clevel, runs, bad, want[0]), flush=True)
    assert bad == 0
print("soak ok")
