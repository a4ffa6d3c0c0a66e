"""ReCoDeWriter.run with and without validation frames (validation_frame_gap 10), interleaved passes on one box, plus what a bare tmpfs write
of the side file's bytes costs on N threads.  usage: validation_gap_rate.py [nframes] [passes]"""
import os, sys, time, tempfile, shutil
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrecode_amd import _lib as hip
from pyrecode_amd.params import InputParams
from pyrecode_amd.recode_writer import ReCoDeWriter

nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 512
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ny = nx = 4096
N = ny * nx
L = hip.lib()
dark = torch.empty(N, dtype=torch.int16, device="cuda")
stack = torch.empty((64, N), dtype=torch.int16, device="cuda")
hip.check(L.rc_synth_dark(0, 7, N, dark.data_ptr()))
hip.check(L.rc_synth_frames(0, 7, 0, 64, N, 10000, dark.data_ptr(), stack.data_ptr()))
h = stack.cpu().numpy().view(np.uint16).reshape(64, ny, nx)
data = np.concatenate([h] * (nfr // 64))
dark_h = dark.cpu().numpy().view(np.uint16).reshape(ny, nx)
del stack
ip = InputParams()
ip._param_map.update(dict(reduction_level=1, rc_operation_mode=1, calibration_threshold_epsilon=0, target_bit_depth=16, source_bit_depth=16,
                          num_cols=nx, num_rows=ny, num_frames=data.shape[0], frame_offset=0, num_calibration_frames=1,
                          calibration_frame_offset=0, keep_part_files=1, num_threads=1, l2_statistics=0, l4_centroiding=0,
                          compression_scheme=2, compression_level=1, source_file_type=0, source_header_length=0,
                          keep_calibration_data=0, calibration_file_type=0, source_data_type=0, target_data_type=0))
out_dir = tempfile.mkdtemp(dir='/dev/shm')
try:
    sweep = [int(v) for v in os.environ.get("VAL_THREADS_SWEEP", "").split(",") if v]
    for nthr in sweep:        # validation writer threads, interleaved with the plain run
        r0, r1 = [], []
        os.environ["RC_WRITER_VAL_THREADS"] = str(nthr)
        for rep in range(passes + 1):
            for gap, acc in ((-1, r0), (10, r1)):
                w = ReCoDeWriter('stack.bin', dark_data=dark_h, output_directory=out_dir, input_params=ip, mode='batch', node_id=0, batch_size=32,
                                 validation_frame_gap=gap)
                w.start()
                t0 = time.perf_counter()
                w.run(data)
                dt = time.perf_counter() - t0
                w.close()
                if rep:
                    acc.append(data.shape[0] / dt)
        r0.sort(); r1.sort()
        print("validation writer threads %d: plain %.1f  gap 10 %.1f (min %.1f max %.1f)  ratio %.3f" % (nthr, r0[len(r0) // 2], r1[len(r1) // 2], r1[0], r1[-1], r1[len(r1) // 2] / r0[len(r0) // 2]), flush=True)
    os.environ.pop("RC_WRITER_VAL_THREADS", None)
    res = {-1: [], 10: []}
    for rep in range(passes + 1):
        for gap in (-1, 10):
            w = ReCoDeWriter('stack.bin', dark_data=dark_h, output_directory=out_dir, input_params=ip, mode='batch', node_id=0, batch_size=32,
                             validation_frame_gap=gap)
            w.start()
            t0 = time.perf_counter()
            w.run(data)
            dt = time.perf_counter() - t0
            w.close()
            if rep:
                res[gap].append(data.shape[0] / dt)
    for gap in (-1, 10):
        r = sorted(res[gap])
        print("gap %3d: frames/s median %.1f  min %.1f  max %.1f" % (gap, r[len(r) // 2], r[0], r[-1]))
    print("ratio of medians: %.3f" % (sorted(res[10])[len(res[10]) // 2] / sorted(res[-1])[len(res[-1]) // 2]))
    # the side file alone: the same bytes to a fresh tmpfs file, 8 MB pieces on n threads
    nval = (data.shape[0] + 9) // 10
    for nthr in (1, 2, 4, 8):
        path = os.path.join(out_dir, 'probe.bin')
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        jobs = [(memoryview(data[10 * k]).cast('B')[lo:lo + (8 << 20)], k * data[0].nbytes + lo) for k in range(nval) for lo in range(0, data[0].nbytes, 8 << 20)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            list(ex.map(lambda j: os.pwrite(fd, j[0], j[1]), jobs))
        dt = time.perf_counter() - t0
        os.close(fd)
        os.remove(path)
        print("side file alone, %d threads: %.2f GB in %.3f s = %.2f GB/s" % (nthr, nval * data[0].nbytes / 1e9, dt, nval * data[0].nbytes / 1e9 / dt))
finally:
    shutil.rmtree(out_dir, ignore_errors=True)
