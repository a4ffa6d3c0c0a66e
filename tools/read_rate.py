"""File-inclusive rate of the reader: a merged .rc1 file on tmpfs -> ReCoDeReader.iter_frames_triplets (two batches in flight) -> triplets
in page-locked host memory; next to it get_frames_triplets batch by batch.  usage: read_rate.py [nframes] [scheme] [batch]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip
from pyrecode_amd.params import InputParams
from pyrecode_amd.recode_writer import ReCoDeWriter
from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts

nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scheme = int(sys.argv[2]) if len(sys.argv) > 2 else 1
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ny = nx = 4096
N = ny * nx
L = hip.lib()
dark = torch.empty(N, dtype=torch.int16, device="cuda")
stack = torch.empty((64, N), dtype=torch.int16, device="cuda")
hip.check(L.rc_synth_dark(0, 7, N, dark.data_ptr()))
hip.check(L.rc_synth_frames(0, 7, 0, 64, N, 10000, dark.data_ptr(), stack.data_ptr()))
h = stack.cpu().numpy().view(np.uint16).reshape(64, ny, nx)
data = np.concatenate([h] * (nfr // 64)) if nfr > 64 else h[:nfr].copy()
dark_h = dark.cpu().numpy().view(np.uint16).reshape(ny, nx)
del stack
ip = InputParams()
ip._param_map.update(dict(reduction_level=1, rc_operation_mode=1, calibration_threshold_epsilon=0, target_bit_depth=16, source_bit_depth=16,
                          num_cols=nx, num_rows=ny, num_frames=data.shape[0], frame_offset=0, num_calibration_frames=1,
                          calibration_frame_offset=0, keep_part_files=1, num_threads=1, l2_statistics=0, l4_centroiding=0,
                          compression_scheme=scheme, compression_level=1, source_file_type=0, source_header_length=0,
                          keep_calibration_data=0, calibration_file_type=0, source_data_type=0, target_data_type=0))
out_dir = tempfile.mkdtemp(dir='/dev/shm')
try:
    w = ReCoDeWriter('stack.bin', dark_data=dark_h, output_directory=out_dir, input_params=ip, mode='batch', node_id=0, batch_size=32)
    w.start()
    w.run(data)
    w.close()
    merge_parts(out_dir, 'stack.rc1', 1)
    path = os.path.join(out_dir, 'stack.rc1')
    size = os.path.getsize(path)
    rd = ReCoDeReader(path)
    rd.open(print_header=False)
    acc = {}
    if os.environ.get("RC_READ_PROFILE"):      # where the streaming form's host time goes
        def timed(name, fn):
            def w(*a, **k):
                t = time.perf_counter()
                r = fn(*a, **k)
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
                return r
            return w
        rd._read_into = timed("file read", rd._read_into)
        real = hip.lib()

        class Proxy:
            def __getattr__(self, k):
                f = getattr(real, k)
                return timed(k, f) if k.startswith("rc_expand") else f
        import pyrecode_amd.recode_reader as rr
        rr._lib = type("L", (), {k: getattr(hip, k) for k in dir(hip) if not k.startswith("__")})
        rr._lib.lib = staticmethod(lambda: Proxy())
    for rep in range(3):
        t0 = time.perf_counter()
        total = 0
        for a, pfx, trip in rd.iter_frames_triplets(0, nfr, batch=batch):
            total += int(pfx[-1])
        dt = time.perf_counter() - t0
        print("[read, file-inclusive] streaming scheme=%d rep=%d: %d frames 4096x4096 from a %.0f MB file on tmpfs -> triplets in page-locked host memory: "
              "%.0f frames/s (%d set pixels)" % (scheme, rep, nfr, size / 1e6, nfr / dt, total))
        if acc:
            print("    of %.1f ms: %s" % (dt * 1e3, ", ".join("%s %.1f" % (k, v * 1e3) for k, v in acc.items())))
            acc.clear()
    for rep in range(2):
        t0 = time.perf_counter()
        total = 0
        for a in range(0, nfr, batch):
            pfx, trip = rd.get_frames_triplets(a, min(batch, nfr - a))
            total += int(pfx[-1])
        dt = time.perf_counter() - t0
        print("[read, file-inclusive] one call per batch scheme=%d rep=%d: %.0f frames/s (%d set pixels)" % (scheme, rep, nfr / dt, total))
    for fused in (True, False):
        rd._no_fused_frame = not fused
        t0 = time.perf_counter()
        k = min(nfr, 64)
        tot = 0
        for z in range(k):
            tot += rd.get_frame(z)[z]['data'].nnz
        dt = time.perf_counter() - t0
        print("[read, file-inclusive] get_frame one by one (%s): %.0f frames/s (%d set pixels)" % ("one device call per frame" if fused else "reference's three steps", k / dt, tot))
    rd.close()
finally:
    shutil.rmtree(out_dir, ignore_errors=True)
