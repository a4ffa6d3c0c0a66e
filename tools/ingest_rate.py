"""Ingest-inclusive rate of ReCoDeWriter: host frames (numpy, in RAM) -> part file on tmpfs.  usage: ingest_rate.py [nframes] [scheme] [mode]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip
from pyrecode_amd.params import InputParams
from pyrecode_amd.recode_writer import ReCoDeWriter

nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 128
scheme = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = sys.argv[3] if len(sys.argv) > 3 else 'auto'
os.environ['RC_WRITER_PIN'] = mode
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
    for rep in range(2):
        w = ReCoDeWriter('stack.bin', dark_data=dark_h, output_directory=out_dir, input_params=ip, mode='batch', node_id=0, batch_size=32)
        w.start()
        t0 = time.perf_counter()
        m = w.run(data)
        dt = time.perf_counter() - t0
        w.close()
        size = os.path.getsize(os.path.join(out_dir, 'stack.rc1_part000'))
        print("[ingest-inclusive] mode=%s scheme=%d rep=%d: %d frames 4096x4096 host RAM -> part file on tmpfs: %.1f frames/s (%.2f GB/s in), part file %.1f MB"
              % (mode, scheme, rep, data.shape[0], data.shape[0] / dt, data.nbytes / dt / 1e9, size / 1e6))
finally:
    shutil.rmtree(out_dir, ignore_errors=True)
