import sys, os, struct
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pyrecode_amd import _lib as hip
from pyrecode_amd.recode_compressors import _zstd_host_decompress
L = hip.lib()
ny = nx = 4096; N = ny*nx; B = 8
for scheme, clevel, d in ((1, 0, 16), (1, 1, 16), (1, 1, 12), (2, 1, 16)):
    dark = torch.empty(N, dtype=torch.int16, device="cuda"); fr = torch.empty((B, N), dtype=torch.int16, device="cuda")
    hip.check(L.rc_synth_dark(0, 1, N, dark.data_ptr())); hip.check(L.rc_synth_frames(0, 1, 0, B, N, 10000, dark.data_ptr(), fr.data_ptr()))
    ctx = hip.ReduceContext(nx, ny, d, 1, 1, scheme, clevel, 0, max_batch=B)
    ctx.set_threshold(dark.data_ptr()); ctx.keep_binary_maps(False)
    frames_h = fr.cpu().numpy().view(np.uint16); dark_h = dark.cpu().numpy().view(np.uint16)
    out, rec, md = ctx.reduce_compress_batch(frames_h, first_frame_id=0)
    tot = 0
    for z in range(B):
        r = out[int(rec[z]):int(rec[z+1])].tobytes()
        fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        if scheme == 1:
            binary = frames_h[z] > dark_h
            bm = np.packbits(binary, bitorder='little').tobytes()
            assert _zstd_host_decompress(r[16:16+cb]) == bm, "bitmap frame %d" % z
            got = _zstd_host_decompress(r[16+cb:])
            assert len(got) == npk
    print("scheme %d clevel %d d %d: record %.0f B/frame, bitmap %.4f of raw, pix %.4f of packed" % (scheme, clevel, d, float(rec[B])/B, cb/(N/8), cp/max(npk,1)))
    ctx.close()
