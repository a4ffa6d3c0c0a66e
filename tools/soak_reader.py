"""Development soak: the streaming reader (rc_expand_frames_submit / _wait) over many batches of different content, every result compared
with the one-call form; interleaved with a pipelined writer on the same device.  usage: soak_reader.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyrecode_amd import _lib as hip
L = hip.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ny, nx, d, B = 512, 1024, 12, 8
N = ny * nx
rng = np.random.default_rng(1)
sets = []
for scheme in (1, 2):
    for k in range(4):
        dark = torch.empty(N, dtype=torch.int16, device="cuda")
        stack = torch.empty((B, N), dtype=torch.int16, device="cuda")
        hip.check(L.rc_synth_dark(0, 100 + k, N, dark.data_ptr()))
        hip.check(L.rc_synth_frames(0, 100 + k, 0, B, N, 5000 * (k + 1), dark.data_ptr(), stack.data_ptr()))
        ctx = hip.ReduceContext(nx, ny, d, 1, 1, scheme, 1, 0, max_batch=B)
        ctx.set_dark(dark.data_ptr(), 0)
        out, rec, md = ctx.reduce_compress_batch(stack.cpu().numpy().view(np.uint16).reshape(B, ny, nx), first_frame_id=0)
        ctx.close()
        blobs = [out[int(rec[z]) + 16:int(rec[z + 1])] for z in range(B)]
        tot = sum(b.size for b in blobs)
        pin = hip.PinnedBuffer(tot + 64)
        np.concatenate(blobs, out=pin.array[:tot])
        sizes = np.ascontiguousarray(md[:, :3], dtype=np.uint32)
        cap = int((sizes[:, 2].astype(np.uint64) * 8 // d).sum())
        prefix = np.zeros(B + 1, np.uint64)
        want = np.zeros((cap, 3), np.uint64)
        hip.check(L.rc_expand_frames(nx, ny, d, 1, 1, scheme, hip.ptr(pin.array[:tot]), hip.ptr(sizes), B, hip.ptr(prefix), hip.ptr(want), cap))
        sets.append((scheme, pin, tot, sizes, cap, prefix.copy(), want[:int(prefix[B])].copy()))
outs = [torch.zeros((max(s[4] for s in sets), 3), dtype=torch.int64, device="cuda") for _ in range(2)]
prefix = np.zeros(B + 1, np.uint64)
pending = [None, None]
bad = 0
t0 = time.time()
for i in range(iters + 1):
    slot = i & 1
    if i < iters:
        k = int(rng.integers(len(sets)))
        scheme, pin, tot, sizes, cap = sets[k][:5]
        hip.check(L.rc_expand_frames_submit(slot, nx, ny, d, 1, 1, scheme, hip.ptr(pin.array[:tot]), hip.ptr(sizes), B, outs[slot].data_ptr(), cap))
        pending[slot] = k
    prev = (i - 1) & 1
    if i >= 1 and pending[prev] is not None:
        hip.check(L.rc_expand_frames_wait(prev, hip.ptr(prefix)))
        k = pending[prev]
        pending[prev] = None
        n = int(prefix[B])
        ok = np.array_equal(prefix, sets[k][5]) and np.array_equal(outs[prev][:n].cpu().numpy().view(np.uint64), sets[k][6])
        bad += 0 if ok else 1
    if i % 100 == 0:
        print("iteration", i, "mismatches", bad, flush=True)
print("soak done: %d batches, %d mismatches, %.1f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
