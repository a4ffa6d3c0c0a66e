"""ctypes binding of librecode_hip.so (C ABI: include/recode_hip.h).

The library is hand-written HIP for gfx950 and has no CPU path; this module fails loudly when the shared object
is missing or when a compute call finds no GPU.  Nothing here imports the oracle.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RC_LIB_PATH") or os.path.join(_HERE, "librecode_hip.so")   # (RC_LIB_PATH: A/B runs of two builds, tools/ab.sh)

RC_OK = 0
RC_ERR_BAD_ARG = -1
RC_ERR_OUT_TOO_SMALL = -2
RC_ERR_DEVICE = -3
RC_ERR_UNSUPPORTED = -4
RC_ERR_RECORD_TOO_LARGE = -5
RC_ERR_CORRUPT = -6
RC_ERR_WORKSPACE = -7
RC_SCHEME_ZLIB_DEVICE = 0x100   # compression_scheme 0 records from the device's own DEFLATE encoder (include/recode_hip.h)

# every symbol include/recode_hip.h declares: (restype, argtypes)
_u8p, _u16p, _u32p, _u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # raw addresses: host or device
SIGNATURES = {
    "rc_abi_version": (C.c_int, []),
    "rc_strerror": (C.c_char_p, [C.c_int]),
    "rc_last_error": (C.c_char_p, []),
    "rc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rc_scheme_on_device": (C.c_int, [C.c_uint32]),
    "rc_ctx_create": (C.c_void_p, [C.c_uint32] * 7 + [C.c_int, C.c_uint32, C.POINTER(C.c_int)]),
    "rc_ctx_destroy": (C.c_int, [C.c_void_p]),
    "rc_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_ctx_set_source_bytes": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rc_ctx_source_bytes": (C.c_uint32, [C.c_void_p]),
    "rc_set_dark": (C.c_int, [C.c_void_p, _u16p, C.c_int64]),
    "rc_set_threshold": (C.c_int, [C.c_void_p, _u16p]),
    "rc_out_capacity": (C.c_uint64, [C.c_void_p, C.c_uint32]),
    "rc_md_fields": (C.c_uint32, [C.c_void_p]),
    "rc_reduce_compress_batch": (C.c_int, [C.c_void_p, _u16p, C.c_uint32, C.c_uint32, _u8p, C.c_uint64, _u64p, _u32p]),
    "rc_reduce_compress_batch_async": (C.c_int, [C.c_void_p, _u16p, C.c_uint32, C.c_uint32, _u8p, C.c_uint64, _u64p, _u32p]),
    "rc_ctx_sync": (C.c_int, [C.c_void_p]),
    "rc_ctx_set_pipelined": (C.c_int, [C.c_void_p, C.c_int]),
    "rc_ctx_wait_results": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_ctx_refit_model": (C.c_int, [C.c_void_p]),
    "rc_get_binary_map": (C.c_int, [C.c_void_p, C.c_uint32, _u8p]),
    "rc_ctx_keep_binary_maps": (C.c_int, [C.c_void_p, C.c_int]),
    "rc_ctx_set_l2_statistics": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rc_get_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rc_ctx_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "rc_ctx_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rc_host_alloc": (C.c_void_p, [C.c_uint64]),
    "rc_host_free": (C.c_int, [C.c_void_p]),
    "rc_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rc_host_unregister": (C.c_int, [C.c_void_p]),
    "rc_pipe_submit": (C.c_int, [C.c_void_p, C.c_uint32, _u16p, C.c_uint32, C.c_uint32]),
    "rc_pipe_input_done": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rc_pipe_result": (C.c_int, [C.c_void_p, C.c_uint32, _u64p, _u32p, C.POINTER(C.c_uint64)]),
    "rc_pipe_fetch": (C.c_int, [C.c_void_p, C.c_uint32, _u8p, C.c_uint64]),
    "rc_pipe_fetch_wait": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rc_ctx_set_validation": (C.c_int, [C.c_void_p] + [C.c_uint32] * 5),
    "rc_pipe_validation": (C.c_int, [C.c_void_p, C.c_uint32, _u32p]),
    "rc_compress": (C.c_int, [C.c_uint32, C.c_uint32, _u8p, C.c_uint64, _u8p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rc_decompress": (C.c_int, [C.c_uint32, _u8p, C.c_uint64, _u8p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rc_compress_bound": (C.c_uint64, [C.c_uint32, C.c_uint64]),
    "rc_unpack_frame_sparse": (C.c_int64, [C.c_uint32, C.c_uint32, C.c_uint32, _u8p, _u8p, C.c_uint64, _u64p, C.c_uint64, C.c_uint32]),
    "rc_expand_frames": (C.c_int, [C.c_uint32] * 6 + [_u8p, _u32p, C.c_uint32, _u64p, _u64p, C.c_uint64]),
    "rc_expand_frames_submit": (C.c_int, [C.c_uint32] * 7 + [_u8p, _u32p, C.c_uint32, _u64p, C.c_uint64]),
    "rc_expand_frames_wait": (C.c_int, [C.c_uint32, _u64p]),
    "rc_expand_frames_coo": (C.c_int, [C.c_uint32] * 6 + [_u8p, _u32p, C.c_uint32, _u64p, C.c_void_p, C.c_uint64]),
    "rc_expand_frames_coo_submit": (C.c_int, [C.c_uint32] * 7 + [_u8p, _u32p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "rc_host_decoder_available": (C.c_int, [C.c_uint32]),
    "rc_host_decode_streams": (C.c_int, [C.c_uint32, _u8p, _u8p, _u64p, C.c_uint32, C.c_uint32]),
    "rc_split_triplets": (C.c_int, [_u64p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "rc_bit_pack": (C.c_int, [_u16p, C.c_uint64, C.c_uint32, _u8p, C.c_uint64]),
    "rc_bit_unpack": (C.c_int, [_u8p, C.c_uint64, C.c_uint64, C.c_uint32, _u64p]),
    "rc_synth_dark": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, _u16p]),
    "rc_synth_frames": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _u16p, _u16p]),
    "rc_synth_frames_clustered": (C.c_int, [C.c_int] + [C.c_uint32] * 6 + [_u16p, _u16p]),
}

_lib = None


class RecodeHipError(RuntimeError):
    pass


def lib():
    """Load librecode_hip.so (once).  If torch is going to be used in this process it must be imported first: both link
    the HIP runtime by SONAME and exactly one copy may be live ("one HIP runtime per process")."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RecodeHipError(
                "librecode_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C pyrecode_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        if "torch" not in sys.modules:
            try:  # keep a single HIP runtime in the process: let torch (if installed) load its own first
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == header / library mismatch
            fn.restype, fn.argtypes = res, args
        if L.rc_abi_version() != 3:
            raise RecodeHipError("librecode_hip.so ABI version mismatch")
        _lib = L
    return _lib


def last_error():
    return (lib().rc_last_error() or b"").decode()


def check(status, what=""):
    """Map an rc_status to the exception type the reference raises at the same point."""
    if status >= 0:
        return status
    msg = "%s%s: %s" % (what + ": " if what else "", lib().rc_strerror(status).decode(), last_error())
    if status == RC_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if status in (RC_ERR_BAD_ARG, RC_ERR_RECORD_TOO_LARGE, RC_ERR_OUT_TOO_SMALL, RC_ERR_CORRUPT):
        raise ValueError(msg)
    if status == RC_ERR_WORKSPACE:
        raise MemoryError(msg)
    raise RecodeHipError(msg)


def device_count():
    n = C.c_int(0)
    check(lib().rc_device_count(C.byref(n)))
    return n.value


def ptr(a):
    """Address of a numpy array's data, or an int address (e.g. torch.Tensor.data_ptr()) passed through."""
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if a is None:
        return None
    return int(a)


PIPE_SLOTS = 3


class PinnedBuffer:
    """Page-locked host memory (rc_host_alloc) seen as a numpy uint8 array: the staging the streaming form copies from / to
    without the HIP runtime's intermediate copy."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._p = lib().rc_host_alloc(self.nbytes)
        if not self._p:
            raise RecodeHipError("rc_host_alloc(%d): %s" % (self.nbytes, last_error()))
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self._p))[:self.nbytes]

    def close(self):
        p, self._p = getattr(self, "_p", None), None
        if p and _lib is not None:
            self.array = None
            _lib.rc_host_free(p)

    __del__ = close


class ReduceContext:
    """One writer's device state: rc_ctx_create .. rc_ctx_destroy.  Mirrors what ReCoDeWriter.start() sets up
    (reference recode_writer.py:212-230) plus the threshold frame of __init__ (:126-137)."""

    def __init__(self, nx, ny, src_bit_depth, reduction_level=1, op_mode=1, scheme=2, clevel=1, device_id=0, max_batch=16, src_dtype=np.uint16,
                 device_zlib=False):
        """src_dtype: numpy dtype of the frames and the dark frame - uint16, or uint8 (source_bit_depth <= 8: rc_ctx_set_source_bytes)
        device_zlib: compression_scheme 0 encoded by the device's DEFLATE encoder (valid zlib streams, not stock zlib's bytes)"""
        if device_zlib and scheme == 0 and op_mode == 1:
            scheme = RC_SCHEME_ZLIB_DEVICE
        st = C.c_int(0)
        self.src_dtype = np.dtype(src_dtype)
        if self.src_dtype not in (np.dtype(np.uint16), np.dtype(np.uint8), np.dtype(np.uint32)):
            raise NotImplementedError("source dtype %s: the device path takes unsigned integer frames (uint8, uint16, uint32)" % self.src_dtype)
        self._h = lib().rc_ctx_create(nx, ny, src_bit_depth, reduction_level, op_mode, scheme, clevel, device_id, max_batch,
                                      C.byref(st))
        if not self._h:
            check(st.value, "rc_ctx_create")
        self.nx, self.ny, self.depth, self.level = nx, ny, src_bit_depth, reduction_level
        self.op_mode, self.scheme, self.max_batch, self.device_id = op_mode, scheme, max_batch, device_id
        self.n_pixels = nx * ny
        self.bitmap_bytes = (self.n_pixels + 7) // 8
        self.on_device_codec = bool(op_mode == 1 and lib().rc_scheme_on_device(scheme))
        if self.src_dtype.itemsize != 2:
            st = lib().rc_ctx_set_source_bytes(self._h, self.src_dtype.itemsize)
            if st != RC_OK:
                self.close()
                check(st, "rc_ctx_set_source_bytes")

    @property
    def handle(self):
        return self._h

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:  # _lib is None during interpreter shutdown
            _lib.rc_ctx_destroy(h)

    __del__ = close

    def set_stream(self, hip_stream):
        check(lib().rc_ctx_set_stream(self._h, hip_stream), "rc_ctx_set_stream")

    def set_dark(self, dark, epsilon=0):
        dark = np.ascontiguousarray(dark, dtype=self.src_dtype) if isinstance(dark, np.ndarray) else dark
        check(lib().rc_set_dark(self._h, ptr(dark), int(epsilon)), "rc_set_dark")

    def set_threshold(self, thr):
        if isinstance(thr, np.ndarray):   # uint16 thresholds for uint8 and uint16 sources, uint32 ones for uint32 sources (include/recode_hip.h)
            thr = np.ascontiguousarray(thr, dtype=np.uint32 if self.src_dtype.itemsize == 4 else np.uint16)
        check(lib().rc_set_threshold(self._h, ptr(thr)), "rc_set_threshold")

    def out_capacity(self, n):
        return lib().rc_out_capacity(self._h, n)

    def reduce_compress_batch(self, frames, first_frame_id=0, out=None):
        """frames: uint16[n, ny, nx] (numpy; uint8 for a ctx of uint8 sources).  Returns (out u8 array, rec_offsets u64[n+1], md u32[n,3])."""
        frames = np.ascontiguousarray(frames, dtype=self.src_dtype)
        n = frames.shape[0]
        if out is None:
            out = np.empty(self.out_capacity(n), np.uint8)
        rec = np.zeros(n + 1, np.uint64)
        md = np.zeros((n, 3), np.uint32)
        check(lib().rc_reduce_compress_batch(self._h, ptr(frames), n, first_frame_id, ptr(out), out.size, ptr(rec), ptr(md)),
              "rc_reduce_compress_batch")
        return out, rec, md

    def enqueue(self, frames_dev, n, first_frame_id, out_dev, out_cap, rec_dev, md_dev):
        check(lib().rc_reduce_compress_batch_async(self._h, ptr(frames_dev), n, first_frame_id, ptr(out_dev), out_cap,
                                                   ptr(rec_dev), ptr(md_dev)), "rc_reduce_compress_batch_async")

    def sync(self):
        check(lib().rc_ctx_sync(self._h), "rc_ctx_sync")

    def set_pipelined(self, on=True):
        """Let the next batch's reduce kernel overlap this batch's scans / layout / assembly (include/recode_hip.h)."""
        check(lib().rc_ctx_set_pipelined(self._h, 1 if on else 0), "rc_ctx_set_pipelined")

    def wait_results(self, stream_handle=None):
        """Order `stream_handle` (None: the ctx's stream) behind the most recent batch's records."""
        check(lib().rc_ctx_wait_results(self._h, C.c_void_p(stream_handle) if stream_handle else None), "rc_ctx_wait_results")

    # ---- host streaming form (include/recode_hip.h, rc_pipe_*) ---------------------------------------------------------
    def pipe_submit(self, slot, frames_host, n, first_frame_id):
        check(lib().rc_pipe_submit(self._h, slot, ptr(frames_host), n, first_frame_id), "rc_pipe_submit")

    def pipe_input_done(self, slot):
        check(lib().rc_pipe_input_done(self._h, slot), "rc_pipe_input_done")

    def pipe_result(self, slot, n):
        """Wait for the slot's batch: (rec_offsets u64[n+1], md u32[n,3], total record bytes)."""
        rec = np.zeros(n + 1, np.uint64)
        md = np.zeros((n, 3), np.uint32)
        total = C.c_uint64(0)
        check(lib().rc_pipe_result(self._h, slot, ptr(rec), ptr(md), C.byref(total)), "rc_pipe_result")
        return rec, md, total.value

    def pipe_fetch(self, slot, dst_host, nbytes):
        check(lib().rc_pipe_fetch(self._h, slot, ptr(dst_host), nbytes), "rc_pipe_fetch")

    def pipe_fetch_wait(self, slot):
        check(lib().rc_pipe_fetch_wait(self._h, slot), "rc_pipe_fetch_wait")

    def set_validation(self, gap, x0=0, y0=0, w=0, h=0):
        """Validation frames on the streaming path: count the ROI's connected components on the device (include/recode_hip.h)."""
        check(lib().rc_ctx_set_validation(self._h, int(gap), int(x0), int(y0), int(w), int(h)), "rc_ctx_set_validation")

    def pipe_validation(self, slot, n):
        """uint32[n]: component count of the ROI per frame of the slot's batch, 0xFFFFFFFF where the frame is no validation frame."""
        counts = np.zeros(n, np.uint32)
        check(lib().rc_pipe_validation(self._h, slot, ptr(counts)), "rc_pipe_validation")
        return counts

    def refit_model(self):
        """zstd, modelled encoder: fit the entropy tables again to the next batch (include/recode_hip.h)."""
        check(lib().rc_ctx_refit_model(self._h), "rc_ctx_refit_model")

    def binary_map(self, i):
        out = np.empty(self.bitmap_bytes, np.uint8)
        check(lib().rc_get_binary_map(self._h, i, ptr(out)), "rc_get_binary_map")
        return out

    def set_l2_statistics(self, code):
        check(lib().rc_ctx_set_l2_statistics(self._h, int(code)), "rc_ctx_set_l2_statistics")

    def keep_binary_maps(self, on=True):
        check(lib().rc_ctx_keep_binary_maps(self._h, 1 if on else 0))

    def set_profiling(self, on=True, every=1):
        """Stage events on the asynchronous path; every = k > 1: around every k-th batch only."""
        check(lib().rc_ctx_set_profiling(self._h, (max(int(every), 1) if on else 0)))

    def profile(self):
        """(sum_ms[5], batches) accumulated by the asynchronous path since set_profiling()."""
        s, n = (C.c_double * 5)(), C.c_uint64(0)
        check(lib().rc_ctx_get_profile(self._h, s, C.byref(n)))
        return list(s), n.value

    def stage_ms(self):
        ms = (C.c_float * 5)()
        check(lib().rc_get_stage_ms(self._h, ms))
        return list(ms)
