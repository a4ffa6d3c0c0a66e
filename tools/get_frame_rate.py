"""Development: rate of the reference's frame-at-a-time calls (get_frame / get_next_frame) on a merged file this library wrote.
usage: get_frame_rate.py [scheme 1|2] [nframes]"""
import os, sys, tempfile, time, shutil, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrecode_amd import synth
from pyrecode_amd.params import InputParams
from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
from pyrecode_amd.recode_writer import ReCoDeWriter

scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ny = nx = 4096
N = ny * nx
dark = synth.dark_frame(3, N)
frames = synth.frames(3, 0, nz, N, 10000, dark)
ip = InputParams()
ip._param_map.update(dict(reduction_level=1, rc_operation_mode=1, calibration_threshold_epsilon=0, target_bit_depth=16, source_bit_depth=16,
                          num_cols=nx, num_rows=ny, num_frames=nz, frame_offset=0, num_calibration_frames=1, calibration_frame_offset=0,
                          keep_part_files=1, num_threads=1, l2_statistics=0, l4_centroiding=0, compression_scheme=scheme, compression_level=1,
                          source_file_type=0, source_header_length=0, keep_calibration_data=0, calibration_file_type=0, source_data_type=0,
                          target_data_type=0))
tmp = tempfile.mkdtemp(dir="/dev/shm")
w = ReCoDeWriter("own.bin", dark_data=dark.reshape(ny, nx), output_directory=tmp, input_params=ip, mode="batch", node_id=0)
w.start(); w.run(frames.reshape(nz, ny, nx)); w.close()
merge_parts(tmp, "own.rc1", 1)
rd = ReCoDeReader(os.path.join(tmp, "own.rc1"), is_intermediate=False)
rd.open(print_header=False)
rd.get_frame(0)
for rep in range(2):
    t0 = time.perf_counter()
    nnz = 0
    for z in range(nz):
        nnz += rd.get_frame(z)[z]["data"].nnz
    dt = time.perf_counter() - t0
print("[get_frame] scheme %d, %d frames 4096x4096 1 %%: %.0f frames/s (%.2f ms per call), nnz ok: %s, served by read-ahead: %d" % (scheme, nz, nz / dt, dt / nz * 1e3, nnz == int((frames > dark).sum()), getattr(rd, "readahead_frames_served", 0)))
rd2 = ReCoDeReader(os.path.join(tmp, "own.rc1"), is_intermediate=False)
rd2.open(print_header=False)
rd2._ra_off = True
rd2.get_frame(0)
t0 = time.perf_counter()
for z in range(nz):
    rd2.get_frame(z)
dt2 = time.perf_counter() - t0
print("[get_frame] the same without read-ahead: %.0f frames/s (%.2f ms per call)" % (nz / dt2, dt2 / nz * 1e3))
rd2.close()
rd3 = ReCoDeReader(os.path.join(tmp, "own.rc1_part000"), is_intermediate=True)
rd3.open(print_header=False)
t0 = time.perf_counter()
k = 0
while rd3.get_next_frame() is not None:
    k += 1
dt3 = time.perf_counter() - t0
print("[get_next_frame] part file, %d frames: %.0f frames/s (%.2f ms per call)" % (k, k / dt3, dt3 / k * 1e3))
rd3.close()
pr = cProfile.Profile(); pr.enable()
for z in range(nz):
    rd.get_frame(z)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
rd.close()
shutil.rmtree(tmp, ignore_errors=True)
