"""ReCoDeWriter: frames in, `.rc<L>_part<NNN>` part file out - the reference's writer API over the HIP hot path.

Keeps the constructor arguments, start()/run()/close(), the per-frame seam `_reduce_compress`, the run-metrics keys, the
output file names and every byte of the part-file layout of reference pyrecode/recode_writer.py (ctor :26-182,
start :184-240, run :292-428, _reduce_compress :430-557, close :589-607).  What changed is where the work happens:
run() hands whole batches of frames to librecode_hip (rc_reduce_compress_batch: threshold, bitmap, compaction,
d-bit pack and - for schemes with a device codec - compression and record assembly on the GPU) and appends the returned
records to the part file.  There is no host implementation of those stages here: without the library or a GPU the
constructor raises.
"""
import math
import os
import struct
import warnings
from datetime import datetime, timedelta
from pathlib import Path

import numpy as np

from . import _lib
from . import recode_compressors as compressors
from .em_reader import emfile
from .misc import effective_cpus, rc_cfg as rc
from .params import InitParams, InputParams
from .recode_header import ReCoDeHeader
from .structures import ReCoDeStructures

_STAGE_KEYS = ('frame_thresholding_and_counting_time', 'frame_binary_image_packing_time',
               'frame_pixel_intensity_packing_time', 'frame_binary_image_compression_time',
               'frame_pixel_intensity_compression_time', 'frame_time')


def _pwrite_all(fd, mv, pos):
    while len(mv):
        k = os.pwrite(fd, mv, pos)
        mv, pos = mv[k:], pos + k


def _read_binary_frames(path, ny, nx, dtype):
    """Raw headerless stack, reference pyrecode/fileutils.py:1-8."""
    a = np.fromfile(path, dtype=dtype)
    return a.reshape((-1, ny, nx))


class ReCoDeWriter:

    def __init__(self, image_filename, dark_data=None, dark_filename='', output_directory='', input_params=None,
                 params_filename='', mode='batch', validation_frame_gap=-1, log_filename='recode.log', run_name='run',
                 verbosity=0, use_c=False, max_count=-1, chunk_time_in_sec=0, node_id=0, buffer_size_in_frames=10.0,
                 device_id=None, batch_size=None, device_zlib=None):
        """Arguments as in the reference (recode_writer.py:29-66).  Three additions, all optional:
        device_id   HIP device ordinal (default: LOCAL_RANK, else node_id modulo the visible GPUs)
        batch_size  frames handed to the GPU per call (default: sized to ~1 GiB of input, at most 64)
        device_zlib compression_scheme 0 only: True = both streams of a record are made by the device's DEFLATE encoder (valid zlib
                    streams that the reference's reader - zlib.decompress, recode_compressors.py:43 - expands to the same bytes; the file
                    header still says scheme 0), False = the reference's own `zlib.compress` call on the host (byte-identical files, two
                    orders of magnitude slower).  Default: the environment's RC_DEVICE_ZLIB (1 / 0), else False.
        use_c is accepted for compatibility; the native path is always the HIP library."""
        self._init_params = InitParams(mode, output_directory, image_filename=image_filename,
                                       calibration_filename=dark_filename, params_filename=params_filename,
                                       validation_frame_gap=validation_frame_gap, log_filename=log_filename,
                                       run_name=run_name, verbosity=verbosity, use_c=use_c)
        if input_params is None:
            self._input_params = InputParams()
            self._input_params.load(Path(self._init_params.params_filename))
        else:
            self._input_params = input_params
        if not self._input_params.validate():
            raise ValueError('Invalid input params')
        ip = self._input_params
        # The reference's Python path takes whatever map_dtype yields for (source_data_type, source_bit_depth) - uint8 up to 8 bits, uint16
        # up to 16, uint32 beyond (misc.py:41-49); only its use_c path is uint16-only (recode_writer.py:85-87).  The device path reads all
        # three; data handed over in another dtype is cast to the source dtype, as there (:352-354).
        if np.dtype(ip.source_numpy_dtype) not in (np.dtype(np.uint16), np.dtype(np.uint8), np.dtype(np.uint32)):
            raise NotImplementedError('source dtype %s (source_data_type %d, source_bit_depth %d): the HIP path takes unsigned integer sources '
                                      '(uint8 / uint16 / uint32 frames); signed and float sources are not implemented on device'
                                      % (np.dtype(ip.source_numpy_dtype).name, ip.source_data_type, ip.source_bit_depth))
        if np.dtype(ip.source_numpy_dtype) == np.dtype(np.uint32) and ip.reduction_level == 2:
            raise NotImplementedError('reduction level 2 is not implemented on device for uint32 sources (source_bit_depth > 16)')
        if ip.reduction_level not in (1, 2, 3):
            raise NotImplementedError('reduction level 4 (centroiding) is not implemented on device '
                                      '(non-functional in the reference as well, SURVEY.md 0.5)')

        self._rc_header = ReCoDeHeader()
        self._rc_header.create(self._init_params, ip, True)
        self._rc_header.set('source_header_length', 1024 if ip.source_file_type in (rc.FILE_TYPE_MRC, rc.FILE_TYPE_SEQ) else 0)
        if verbosity > 0:
            self._rc_header.print()
        if not self._rc_header.validate():
            raise ValueError('Invalid ReCoDe header created')
        self._header = self._rc_header.as_dict()

        if dark_data is None:
            if ip.calibration_file_type == rc.FILE_TYPE_BINARY:
                t = _read_binary_frames(self._init_params.calibration_filename, self._header['ny'], self._header['nx'],
                                        ip.source_numpy_dtype)
                t = np.squeeze(t[0]) if t.ndim > 2 else t
            elif ip.calibration_file_type in (rc.FILE_TYPE_MRC, rc.FILE_TYPE_SEQ):   # reference :109-113
                with emfile(self._init_params.calibration_filename, ip.calibration_file_type) as _t:
                    t = np.array(np.squeeze(_t[0]) if len(_t.shape) > 2 else _t[0])
            else:
                raise NotImplementedError("No implementation available for loading calibration file of type 'Other'")
        else:
            t = dark_data
        if self._header['ny'] != t.shape[0] or self._header['nx'] != t.shape[1]:
            raise RuntimeError('Data and Calibration frames have different shapes')
        self._src_dtype = ip.source_numpy_dtype
        if t.dtype != self._src_dtype:
            # reference :131-137 casts after adding epsilon in the dark frame's own dtype; for an integer epsilon
            # cast-then-add gives the same frame of the source dtype (floor(x + k) == floor(x) + k), so the sum stays on the device
            warnings.warn('Calibration data type not same as source. Attempting to cast.')
            t = t.astype(self._src_dtype)
        self._calibration_frame = t  # thr = dark + epsilon is formed on the device (rc_set_dark, reference :126-127)

        self._node_id = node_id
        self._device_id = device_id
        self._batch_size = batch_size
        if device_zlib is None:
            device_zlib = os.environ.get('RC_DEVICE_ZLIB', '0') not in ('', '0')
        self._device_zlib = bool(device_zlib) and ip.compression_scheme == 0 and ip.rc_operation_mode == 1 \
            and np.dtype(ip.source_numpy_dtype) != np.dtype(np.uint32)   # (uint32 sources: the host call, include/recode_hip.h)
        self._buffer_size_in_frames = buffer_size_in_frames
        self._structures = ReCoDeStructures(self._header)
        self._intermediate_file_name = self._intermediate_file = None
        self._validation_file_name = self._validation_file = None
        self._ctx = None
        self._is_first_chunk = True
        self._chunk_offset = self._num_frames_in_part = None
        self._frame_buffer = None
        self._vc_roi = {'x_start': None, 'y_start': None, 'nx': None, 'ny': None}
        self._vc_n_pixels = None
        self._vc_dose_rate = 0.0
        self._compressor_context = None
        self._pin_in = self._pin_out = None
        self._pin_mode = os.environ.get('RC_WRITER_PIN', 'auto')   # auto | register | stage
        # staging copies: a thread moves 5-8 GB/s from pageable memory; the link takes 57 (measured on 2 x 64 cores: 6 threads 46.9 GB/s
        # end to end, 24 threads 51.2)
        # sized by the cores the cgroup really grants (the GPU box shows 256 and grants 16: 24 copy threads next to the stager, the writer
        # thread and the library's workers spent the quota and throttled the whole pipeline)
        self._copy_threads = int(os.environ.get('RC_WRITER_COPY_THREADS', str(max(2, min(24, effective_cpus()[0] // 2)))))
        if ip.compression_scheme == 1 and not _lib.lib().rc_scheme_on_device(1):
            import zstandard as zstd
            self._compressor_context = zstd.ZstdCompressor(level=ip.compression_level, write_content_size=False)

    # ---------------------------------------------------------------------------------------------------------
    def _pick_device(self):
        if self._device_id is not None:
            return int(self._device_id)
        n = _lib.device_count()
        if n == 0:
            raise _lib.RecodeHipError('no GPU visible: ReCoDeWriter has no CPU path')
        if 'LOCAL_RANK' in os.environ:
            return int(os.environ['LOCAL_RANK']) % n
        return self._node_id % n

    def start(self):
        """Create the part file, write its header, allocate host buffers and the device context."""
        ip, init = self._input_params, self._init_params
        base = Path(init.image_filename).stem if init.mode == 'batch' else init.run_name
        self._intermediate_file_name = os.path.join(
            init.output_directory, '%s.rc%d_part%03d' % (base, ip.reduction_level, self._node_id))
        self._intermediate_file = open(self._intermediate_file_name, 'wb')
        self._rc_header.serialize_to(self._intermediate_file)
        self._intermediate_file.flush()
        if init.validation_frame_gap > 0:
            self._validation_file_name = os.path.join(
                init.output_directory, '%s_part%03d_validation_frames.bin' % (base, self._node_id))
            self._validation_file = open(self._validation_file_name, 'wb')
            self._val_pos = 0

        nx, ny = int(self._header['nx']), int(self._header['ny'])
        self._frame_sz = nx * ny * np.dtype(self._src_dtype).itemsize
        self._frame_buffer = bytearray(self._frame_sz)
        self._n_bytes_in_binary_image = math.ceil(nx * ny / 8)
        self._buffer_sz = int(np.ceil(self._frame_sz * self._buffer_size_in_frames))
        self._rct_buffer = bytearray()
        if self._batch_size is None:
            self._batch_size = int(max(1, min(64, (1 << 30) // self._frame_sz)))
        self._ctx = _lib.ReduceContext(nx, ny, ip.source_bit_depth, ip.reduction_level, ip.rc_operation_mode,
                                       ip.compression_scheme, ip.compression_level, self._pick_device(), self._batch_size,
                                       src_dtype=self._src_dtype, device_zlib=self._device_zlib)
        self._ctx.set_dark(np.ascontiguousarray(self._calibration_frame), ip.calibration_threshold_epsilon)
        if ip.reduction_level == 2:
            self._ctx.set_l2_statistics(ip.L2_statistics)  # 0/1 max, 2 sum (reference :358-365)
        # raw binary maps are only written for the per-frame seam (_reduce_compress returns one); validation frames no longer need them
        self._keep_maps = False
        self._ctx.keep_binary_maps(self._keep_maps)
        self._host_compress = ip.rc_operation_mode == 1 and not self._ctx.on_device_codec
        self._out = np.empty(self._ctx.out_capacity(self._batch_size), np.uint8)
        self._chunk_offset = 0
        self._num_frames_in_part = 0
        # per frame written: [frame_id, data bytes, metadata fields...] and where its data starts in the part file - what the
        # direct merge needs (parallel.merge_direct) without reading the part file back
        self._n_md = len(self._structures.standard_frame_metadata_structure_for(ip.reduction_level, ip.rc_operation_mode))
        self._index_rows, self._index_pos = [], []
        self._file_pos = None
        self._vc_roi['nx'], self._vc_roi['ny'] = min(nx, 128), min(ny, 128)
        self._vc_roi['x_start'] = math.floor((nx - self._vc_roi['nx']) / 2.0)
        self._vc_roi['y_start'] = math.floor((ny - self._vc_roi['ny']) / 2.0)
        self._vc_n_pixels = self._vc_roi['nx'] * self._vc_roi['ny']
        if init.validation_frame_gap > 0:
            self._ctx.set_validation(init.validation_frame_gap, self._vc_roi['x_start'], self._vc_roi['y_start'],
                                     self._vc_roi['nx'], self._vc_roi['ny'])
        self._alloc_staging()

    def _do_sanity_checks(self, is_first_chunk, data=None):
        ip = self._input_params
        if data is None:
            if ip.source_file_type == rc.FILE_TYPE_BINARY:
                self._source = None
                self._source_shape = (self._header['nz'], self._header['ny'], self._header['nx'])
            elif ip.source_file_type in (rc.FILE_TYPE_MRC, rc.FILE_TYPE_SEQ):   # reference :250-267
                with emfile(self._init_params.image_filename, ip.source_file_type) as src:
                    self._source = None
                    self._source_shape = tuple(int(v) for v in src.shape)
                    if is_first_chunk:  # the 1024 bytes announced by source_header_length
                        src.serialize_header(self._intermediate_file)
                        self._intermediate_file.flush()
            else:
                raise NotImplementedError("No implementation available for loading calibration file of type 'Other'")
        else:
            self._source = data
            self._source_shape = data.shape
        if self._source_shape[1] != self._header['ny']:
            raise RuntimeError('Expected height does not match height in source file')
        if self._source_shape[2] != self._header['nx']:
            raise RuntimeError('Expected width does not match width in source file')
        if ip.num_frames == -1:
            self._header['nz'] = self._source_shape[0]
        elif ip.num_frames > self._source_shape[0]:
            raise RuntimeError('Number of frames requested in config file is larger than available in source file')
        else:
            self._header['nz'] = ip.num_frames

    # ---------------------------------------------------------------------------------------------------------
    def run(self, data=None):
        """Process this node's contiguous block of the chunk (reference :311-350) and append its records to the part file."""
        run_metrics = {}
        ip, init = self._input_params, self._init_params
        self._do_sanity_checks(self._is_first_chunk, data)
        self._is_first_chunk = False
        if init.mode == 'batch':
            n_frames_in_chunk = ip.nz
        elif init.mode == 'stream':
            n_frames_in_chunk = self._source_shape[0]
        else:
            raise ValueError("Invalid input params: mode. Can be 'batch' or 'stream'.")
        n_frames_per_thread = int(math.ceil(n_frames_in_chunk / float(ip.num_threads)))
        frame_offset = self._node_id * n_frames_per_thread
        available_frames = min(n_frames_per_thread, max(n_frames_in_chunk - frame_offset, 0))

        stt = datetime.now()
        if data is None and ip.source_file_type in (rc.FILE_TYPE_MRC, rc.FILE_TYPE_SEQ):   # reference :329-348
            with emfile(init.image_filename, ip.source_file_type) as f:
                try:  # the header's frame count may overstate what the file holds
                    data = np.array(f[frame_offset:frame_offset + available_frames])
                except IndexError:
                    frame_list = []
                    while len(frame_list) < available_frames:
                        try:
                            frame_list.append(np.squeeze(f[frame_offset + len(frame_list)]))
                        except IndexError:
                            break
                    data = np.asarray(frame_list).reshape((-1, self._header['ny'], self._header['nx']))
                    available_frames = data.shape[0]
        elif data is None:
            stack = _read_binary_frames(init.image_filename, self._header['ny'], self._header['nx'], self._src_dtype)
            data = stack[frame_offset:frame_offset + available_frames]
            available_frames = data.shape[0]
        else:
            data = data[frame_offset:frame_offset + available_frames]
        if data.dtype != self._src_dtype:
            warnings.warn('Source data type either not as specified or does not match params specs. Attempting to cast.')
            data = data.astype(self._src_dtype)
        run_metrics['run_data_read_time'] = datetime.now() - stt

        run_start = datetime.now()
        # One path for every run: batches stream through the device (rc_pipe_*).  Validation frames (reference :402-415) ride
        # along: the raw frame goes to the validation file from the source array, the dose-rate count of its ROI comes from
        # the device with the batch's sizes (rc_ctx_set_validation / rc_pipe_validation).
        for key, value in self._run_streamed(data, available_frames, self._chunk_offset + frame_offset).items():
            run_metrics[key] = value
        self._chunk_offset += n_frames_in_chunk
        self._num_frames_in_part += available_frames
        run_metrics['run_time'] = datetime.now() - run_start
        run_metrics['run_frames'] = available_frames
        return run_metrics

    def _alloc_staging(self):
        """Page-locked staging of the streaming form: three input buffers, PIPE_SLOTS output buffers (grown on demand)."""
        if self._pin_in is None:
            B = self._batch_size
            staged = B * int(self._header['ny']) * int(self._header['nx']) * np.dtype(self._src_dtype).itemsize   # staged in the source dtype (uint16 / uint8)
            self._pin_in = [_lib.PinnedBuffer(staged) for _ in range(3)]
            self._pin_out = [_lib.PinnedBuffer(max(staged // 8, 1 << 20)) for _ in range(_lib.PIPE_SLOTS)]

    def _run_streamed(self, data, n_frames, first_id):
        """The frame loop of the reference (recode_writer.py:383-399) as a stream over batches (rc_pipe_*).  Four things run at
        once: a stager thread fills page-locked staging buffers from the source (a few copy threads; the file pages / the
        caller's array are read exactly once), the GPU reads batch i straight out of its staging buffer and builds its
        records, batch i-1's records travel back into a page-locked buffer, and a writer thread appends batch i-2's records
        to the part file from that buffer - no per-record Python objects (host-compressed schemes excepted: the reference's
        own library call needs the pieces)."""
        from concurrent.futures import ThreadPoolExecutor
        ctx, B = self._ctx, self._batch_size
        frame_bytes = int(self._header['ny']) * int(self._header['nx']) * np.dtype(self._src_dtype).itemsize
        zero = timedelta(0)
        metrics = {k: zero for k in _STAGE_KEYS}
        if n_frames == 0:
            return metrics
        t_run = datetime.now()
        nbatch = -(-n_frames // B)
        slots = _lib.PIPE_SLOTS
        n_in = 3
        self._alloc_staging()
        # RC_WRITER_PIN=register: a stack that is already in RAM is pinned in place instead and read from where it lies
        # (saves the staging copy, costs one page-locking pass over the whole stack up front)
        registered = None
        if self._pin_mode == 'register' and isinstance(data, np.ndarray) and not isinstance(data, np.memmap) \
                and data.flags['C_CONTIGUOUS'] and data.dtype == self._src_dtype:
            try:
                _lib.check(_lib.lib().rc_host_register(data.ctypes.data, data.nbytes), 'rc_host_register')
                registered = data.ctypes.data
            except Exception:
                registered = None
        copy_pool = ThreadPoolExecutor(max_workers=self._copy_threads)
        stager = ThreadPoolExecutor(max_workers=1)
        writer = ThreadPoolExecutor(max_workers=1)
        info = [None] * nbatch        # per batch: [n, rec offsets, md, total]
        staged = [None] * nbatch      # futures of the stager
        written = [None] * nbatch     # futures of the writer
        ny, nx = self._header['ny'], self._header['nx']

        def stage(i):   # (stager thread) batch i -> staging buffer i % n_in, once the batch that last used it has been read
            lo = i * B
            n = min(B, n_frames - lo)
            if registered is not None:
                return data[lo:lo + n]
            if i >= n_in:
                ctx.pipe_input_done((i - n_in) % slots)
            view = self._pin_in[i % n_in].array[:n * frame_bytes].view(self._src_dtype).reshape(n, ny, nx)
            parts = max(1, min(self._copy_threads * 2, n))
            step = -(-n // parts)
            list(copy_pool.map(lambda a: np.copyto(view[a:a + step], data[lo + a:lo + min(a + step, n)], casting='unsafe'),
                               range(0, n, step)))
            return view

        gap = self._init_params.validation_frame_gap
        dose_rates, val_jobs = [], []
        usable, visible = effective_cpus()
        # (ONE writer thread: concurrent pwrite()s into one tmpfs / page-cache file contend - same box, 512 frames, gap 10, rate against the
        # plain run: 1 thread 0.998, 2 threads 0.559, 4 0.527, 8 0.686, 12 0.581 (tools/validation_gap_rate.py, profiles/r04_validation_gap_rate.log);
        # one thread streams the side file at 6-7 GB/s, which a run with every tenth frame a validation frame needs a fifth of)
        val_writer = ThreadPoolExecutor(max_workers=int(os.environ.get('RC_WRITER_VAL_THREADS', '1'))) if gap > 0 else None
        if gap > 0:
            # Validation frames (reference :402-415): WHICH frames go to the side file is known before anything runs - every frame whose id
            # is a multiple of the gap - and where (frame order), so all their writes are queued now, in 8 MB pieces on a thread of their own
            # (pwrite() at its offset, from the source array's memory, releases the GIL; a tmpfs / page-cache write is a memcpy plus page
            # allocation: 1-2 GB/s a thread), and run beside the whole stream instead of trailing the batches they belong to (round 3
            # queued a frame's 32 MB when its batch's records came back: the last batches' frames were written behind the end of the run,
            # and with every tenth frame a validation frame the side file takes three times the bytes of the records).  Only the dose
            # rate - the device's component count of the frame's ROI - still arrives with the batch (append() below).
            fd, piece = self._validation_file.fileno(), 8 << 20
            val_pos_start, n_val_queued = self._val_pos, 0
            for k in range(n_frames):
                if (first_id + k) % gap == 0:
                    n_val_queued += 1
                    mv = memoryview(np.ascontiguousarray(data[k])).cast('B')
                    for lo in range(0, mv.nbytes, piece):
                        val_jobs.append(val_writer.submit(_pwrite_all, fd, mv[lo:lo + piece], self._val_pos + lo))
                    self._val_pos += mv.nbytes
        # a quota'd container is this writer's alone (one rank per GPU box: all of its share); a whole node is shared by its ranks
        host_workers = max(2, min(32, usable if usable < visible else visible // 4))
        host_pool = (ThreadPoolExecutor(max_workers=host_workers)
                     if self._host_compress and int(self._header['compression_scheme']) in (0, 4, 5) else None)

        def append(i):  # (writer thread) batch i's records: page-locked buffer -> part file
            ctx.pipe_fetch_wait(i % slots)
            n, rec, md, total, counts = info[i]
            if counts is not None:   # dose rates of this batch's validation frames, in frame order (reference :402-415; their raw frames
                for k in np.nonzero(counts != 0xFFFFFFFF)[0]:                                    # are on their way to the side file already)
                    self._vc_dose_rate = int(counts[k]) / self._vc_n_pixels
                    dose_rates.append(self._vc_dose_rate)
            buf = self._pin_out[i % slots].array
            pos = self._intermediate_file.tell()
            if self._host_compress:
                # the reference's own library call per stream (zlib, bz2, lzma release the GIL): the batch's records are compressed
                # side by side and written in frame order - same bytes as one after the other
                def one(z):
                    m = {'frame_binary_image_compression_time': zero, 'frame_pixel_intensity_compression_time': zero}
                    return self._host_compress_record(buf[int(rec[z]):int(rec[z + 1])].tobytes(), m), m
                done = list(host_pool.map(one, range(n))) if host_pool is not None else [one(z) for z in range(n)]
                for r, m in done:
                    for key, value in m.items():
                        metrics[key] += value
                    self._note_host_record(pos, r)
                    self._intermediate_file.write(r)
                    pos += len(r)
            else:
                self._note_records(pos, n, rec, md, first_id + i * B)
                self._intermediate_file.write(memoryview(buf)[:total])
            info[i] = None

        try:
            for i in range(min(n_in - 1, nbatch)):
                staged[i] = stager.submit(stage, i)
            for i in range(nbatch + 1):
                if i < nbatch:
                    if i >= slots:
                        written[i - slots].result()      # the slot's previous batch has left its output buffer
                    frames = staged[i].result()
                    n = frames.shape[0]
                    ctx.pipe_submit(i % slots, frames, n, first_id + i * B)
                    info[i] = [n, None, None, 0, None]
                    if i + n_in - 1 < nbatch:
                        staged[i + n_in - 1] = stager.submit(stage, i + n_in - 1)
                if 0 <= i - 1 < nbatch:   # batch i-1 has been computed: sizes known, start copying its records out
                    j = i - 1
                    rec, md, total = ctx.pipe_result(j % slots, info[j][0])
                    info[j][1:] = [rec, md, total, ctx.pipe_validation(j % slots, info[j][0]) if gap > 0 else None]
                    buf = self._pin_out[j % slots]
                    if total > buf.nbytes:
                        buf.close()
                        buf = self._pin_out[j % slots] = _lib.PinnedBuffer(int(total * 1.25))
                    ctx.pipe_fetch(j % slots, buf.array, total)
                    written[j] = writer.submit(append, j)
            for f in written:
                if f is not None:
                    f.result()
            for f in val_jobs:
                f.result()
            if gap > 0 and len(dose_rates) != n_val_queued:   # the host's choice of validation frames (ids % gap) and the device's counts must agree
                raise RuntimeError('validation frames: %d written to the side file, %d dose rates from the device' % (n_val_queued, len(dose_rates)))
        except BaseException:
            # The side file was queued ahead of the batches (above): after a failed run (a record larger than its frame, a device error) it is
            # cut back to the validation frames of the batches whose records DID reach the part file - what the reference's frame-at-a-time
            # loop would have left (recode_writer.py:402-415 writes a validation frame only behind its record).
            if gap > 0:
                for f in [w for w in written if w is not None] + val_jobs:   # what is on its way lands first (appends decide how far the run got)
                    try:
                        f.result()
                    except Exception:   # noqa: BLE001 - already failing
                        pass
                self._val_pos = val_pos_start + len(dose_rates) * frame_bytes
                try:
                    self._validation_file.flush()
                    os.ftruncate(self._validation_file.fileno(), self._val_pos)
                except OSError:
                    pass
            raise
        finally:
            for pool in (stager, writer, copy_pool, val_writer, host_pool):
                if pool is not None:
                    pool.shutdown(wait=True)
            if registered is not None:
                _lib.check(_lib.lib().rc_host_unregister(registered), 'rc_host_unregister')
        self._intermediate_file.flush()
        metrics['frame_time'] = datetime.now() - t_run
        metrics['frame_thresholding_and_counting_time'] = metrics['frame_time']   # one fused, overlapped stream: not separable
        if dose_rates:
            metrics['run_dose_rates'] = dose_rates
        return metrics

    def _reduce_compress_batch(self, frames, first_frame_id):
        """n frames -> list of n record byte strings + metrics summed over the batch (device stage times)."""
        n = frames.shape[0]
        t0 = datetime.now()
        out, rec, md = self._ctx.reduce_compress_batch(frames, first_frame_id, out=self._out)
        ms = self._ctx.stage_ms()
        zero = timedelta(0)
        metrics = {
            'frame_thresholding_and_counting_time': timedelta(milliseconds=ms[0] + ms[2]),
            'frame_binary_image_packing_time': zero,  # fused into the thresholding kernel
            'frame_pixel_intensity_packing_time': timedelta(milliseconds=ms[3]),
            'frame_binary_image_compression_time': timedelta(milliseconds=ms[1]),  # LZ4: fused into the first
            'frame_pixel_intensity_compression_time': zero,  # fused into record assembly
        }
        records = [out[int(rec[i]):int(rec[i + 1])] for i in range(n)]
        if self._host_compress:
            records = [self._host_compress_record(r.tobytes(), metrics) for r in records]
        else:
            records = [r.tobytes() for r in records]
        metrics['frame_time'] = datetime.now() - t0
        return records, metrics

    def _host_compress_record(self, r, metrics):
        """Schemes without a device codec: the GPU delivered the mode-0 record; run the reference's own library call on
        the two streams (recode_writer.py:503-525) and re-frame."""
        h, nb = self._header, self._n_bytes_in_binary_image
        scheme, level = h['compression_scheme'], h['compression_level']
        t0 = datetime.now()
        if h['reduction_level'] in (1, 2):
            fid, npk = struct.unpack_from('<II', r, 0)
            cb = compressors.compress(scheme, level, r[8:8 + nb], self._compressor_context)
            t1 = datetime.now()
            cp = compressors.compress(scheme, level, r[8 + nb:8 + nb + npk], self._compressor_context)
            metrics['frame_pixel_intensity_compression_time'] += datetime.now() - t1
            out = struct.pack('<IIII', fid, len(cb), len(cp), npk) + cb + cp
        else:
            fid, = struct.unpack_from('<I', r, 0)
            cb = compressors.compress(scheme, level, r[4:4 + nb], self._compressor_context)
            t1 = datetime.now()
            out = struct.pack('<II', fid, len(cb)) + cb
        metrics['frame_binary_image_compression_time'] += t1 - t0
        if len(out) > self._frame_sz:
            raise ValueError('Buffer size smaller than compressed data size')
        return out

    def _reduce_compress(self, frame, absolute_frame_index, _statistics=None, _centroiding_scheme=None):
        """The reference's per-frame seam (recode_writer.py:430-557): record bytes land in self._frame_buffer[:length];
        returns (length, metrics, binary_frame)."""
        self._ctx.keep_binary_maps(True)
        records, metrics = self._reduce_compress_batch(np.ascontiguousarray(frame)[None], int(absolute_frame_index))
        rec = records[0]
        self._frame_buffer[:len(rec)] = rec
        nx, ny = int(self._header['nx']), int(self._header['ny'])
        binary = np.unpackbits(self._ctx.binary_map(0), bitorder='little')[:nx * ny].reshape(ny, nx).astype(bool)
        self._ctx.keep_binary_maps(self._keep_maps)
        return len(rec), metrics, binary

    def close(self):
        self._offload_buffer()
        self._rc_header.update('nz', self._num_frames_in_part)
        self._intermediate_file.seek(0)
        self._rc_header.serialize_to(self._intermediate_file)
        self._intermediate_file.close()
        if self._init_params.validation_frame_gap > 0:
            self._validation_file.close()
        for buf in (self._pin_in or []) + (self._pin_out or []):
            buf.close()
        self._pin_in = self._pin_out = None
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def device_id(self):
        """HIP device ordinal this writer's frames are reduced on (valid after start())."""
        if self._ctx is None:
            raise RuntimeError('ReCoDeWriter.device_id(): call start() first')
        return int(self._ctx.device_id)

    def frame_index(self):
        """(rows int64[n, 2 + n_md] = [frame_id, data bytes, metadata...], data offsets int64[n]) of the frames written so far."""
        rows = np.array(self._index_rows, dtype=np.int64).reshape(len(self._index_rows), 2 + self._n_md)
        return rows, np.array(self._index_pos, dtype=np.int64)

    def _note_host_record(self, pos, r):
        """One finished record (bytes) about to be written at file position pos."""
        hdr = 4 + 4 * self._n_md
        vals = struct.unpack_from('<%dI' % (1 + self._n_md), r, 0)
        self._index_rows.append([vals[0], len(r) - hdr] + list(vals[1:]))
        self._index_pos.append(pos + hdr)

    def _note_records(self, pos, n, rec, md, first_id, lengths=None):
        """Bookkeeping for frame_index(): n records starting at file position pos; rec = offsets inside the batch."""
        hdr = 4 + 4 * self._n_md
        for z in range(n):
            lo = int(rec[z])
            size = (int(rec[z + 1]) - lo) if lengths is None else lengths[z]
            self._index_rows.append([first_id + z, size - hdr] + [int(v) for v in md[z][:self._n_md]])
            self._index_pos.append(pos + lo + hdr)

    def _offload_buffer(self):
        self._intermediate_file.write(self._rct_buffer)
        self._intermediate_file.flush()
        self._rct_buffer = bytearray()


def print_run_metrics(run_metrics):
    for key in run_metrics:
        if key.startswith('frame_'):
            print(key, "\t", run_metrics[key] / run_metrics['run_frames'], "\t", run_metrics[key] / run_metrics['frame_time'])
        elif key == 'run_dose_rates':
            print(key, "\t", run_metrics[key], "\t", 'Avg.=', np.mean(run_metrics[key]))
        else:
            print(key, "\t", run_metrics[key])
