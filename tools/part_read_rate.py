"""Development: rate of the batched reader on a PART file (records indexed once, frames gathered per batch) against the merged file.
usage: part_read_rate.py [scheme 1|2] [nframes] [batch]"""
import os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrecode_amd import synth
from pyrecode_amd.params import InputParams
from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
from pyrecode_amd.recode_writer import ReCoDeWriter

scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
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
want = int((frames > dark).sum())
for name, inter in (("own.rc1", False), ("own.rc1_part000", True), ("own.rc1", False), ("own.rc1_part000", True)):
    rd = ReCoDeReader(os.path.join(tmp, name), is_intermediate=inter)
    rd.open(print_header=False)
    t0 = time.perf_counter()
    nfr = rd._batch_frames()
    t_index = time.perf_counter() - t0
    acc, saved, trace = {}, [], []
    def wrap(obj, name):
        fn = getattr(obj, name)
        saved.append((obj, name, fn))
        def timed(*a, **k):
            t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
            trace.append((name[-6:], round((time.perf_counter() - t) * 1e3, 2))); return r
        setattr(obj, name, timed)
    from pyrecode_amd import _lib as _L
    wrap(rd, "_read_batch_into"); wrap(_L.lib(), "rc_expand_frames_submit"); wrap(_L.lib(), "rc_expand_frames_wait")
    for rep in range(2):
        acc.clear()
        t0 = time.perf_counter()
        got = 0
        for a, pre, tr in rd.iter_frames_triplets(batch=batch):
            got += int(pre[-1])
        dt = time.perf_counter() - t0
    assert got == want and nfr == nz
    print("[part-read] %-16s scheme %d, %d frames, batches of %d: %.0f frames/s to host triplets (%s; index %.1f ms)"
          % (name, scheme, nz, batch, nz / dt, rd.last_batch_path, t_index * 1e3))
    print("            of %.1f ms: %s" % (dt * 1e3, ", ".join("%s %.1f ms" % (k, v * 1e3) for k, v in acc.items())))
    for rep in range(2):
        t0 = time.perf_counter()
        got = 0
        for a, pre, (rows, cols, vals) in rd.iter_frames_triplets(batch=batch, coo=True):
            got += rows.shape[0]
        dtc = time.perf_counter() - t0
    assert got == want
    print("[part-read] %-16s the same as COO arrays (10 bytes a set pixel): %.0f frames/s" % (name, nz / dtc))
    print("            calls:", trace)
    for obj, nm, fn in saved:
        setattr(obj, nm, fn)               # (back to the library's own entry points, argtypes and all)
    rd.close()
shutil.rmtree(tmp, ignore_errors=True)
