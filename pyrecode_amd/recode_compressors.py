"""Compressor backend dispatch: compress(scheme, level, data, ctx) / de_compress(scheme, data, ctx).

Same two signatures, scheme codes and NotImplementedError behaviour as reference pyrecode/recode_compressors.py
(:82-120, :40-79, import_checks :123-129).  Schemes with a device codec (rc_scheme_on_device: LZ4 frames today) run
on the GPU through rc_compress / rc_decompress; the others call the same host library the reference calls.  Inside
ReCoDeWriter the device codecs never come through here: the batched operator emits finished records.
"""
import bz2
import ctypes as C
import lzma
import threading
import zlib

import numpy as np

from . import _lib

_compression_scheme_code_map = {0: 'zlib', 1: 'zstandard', 2: 'lz4', 3: 'snappy', 4: 'bzip', 5: 'lzma', 6: 'blosc',
                                7: 'blosc', 8: 'blosc', 9: 'blosc', 10: 'blosc', 11: 'blosc'}
_BLOSC_CNAMES = {6: 'zlib', 7: 'zstd', 8: 'lz4', 9: 'snappy', 10: 'blosclz', 11: 'lz4hc'}


def _optional(name):
    try:
        return __import__(name)
    except ImportError:
        return None


def _on_device(scheme):
    """True when the GPU library ENCODES this scheme itself (LZ4 frames, zstd frames)."""
    return bool(_lib.lib().rc_scheme_on_device(int(scheme)))


_DEVICE_DECODERS = (2, 8)  # LZ4 frames and blosc1-LZ4 chunks are always DEcoded on the GPU; zstd frames when they lie inside the
# device decoder's subset (everything this library writes; rc_decompress says RC_ERR_UNSUPPORTED otherwise), else by the stock library


_HOST_LIBS = {}


def _host_lib(name):
    """The system's libzstd / liblz4 through ctypes, loaded ONCE (ctypes.util.find_library runs ldconfig: tens of milliseconds a call)."""
    if name not in _HOST_LIBS:
        import ctypes.util
        path = ctypes.util.find_library(name)
        L = C.CDLL(path) if path else None
        if L is not None and name == 'zstd':
            L.ZSTD_createDStream.restype = C.c_void_p
            L.ZSTD_freeDStream.argtypes = [C.c_void_p]
            L.ZSTD_decompressStream.restype = C.c_size_t
            L.ZSTD_decompressStream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            L.ZSTD_isError.argtypes = [C.c_size_t]
            L.ZSTD_initDStream.restype = C.c_size_t
            L.ZSTD_initDStream.argtypes = [C.c_void_p]
            L.ZSTD_decompressDCtx.restype = C.c_size_t
            L.ZSTD_decompressDCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        if L is not None and name == 'lz4':
            L.LZ4F_createDecompressionContext.restype = C.c_size_t
            L.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            L.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
            L.LZ4F_decompress.restype = C.c_size_t
            L.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
            L.LZ4F_isError.argtypes = [C.c_size_t]
            L.LZ4F_resetDecompressionContext.restype = None
            L.LZ4F_resetDecompressionContext.argtypes = [C.c_void_p]
        _HOST_LIBS[name] = L
    return _HOST_LIBS[name]


class _ThreadCtx:
    """One decoding context of a stock library per THREAD, kept for the thread's life.  A fresh context per stream costs more than
    its allocation: its 100 KB - 2 MB buffers come from mmap and go back with munmap, and with 16 decoding threads in one process
    those calls (address-space lock, TLB shootdowns) serialise the pool - measured: 8 threads no faster than one."""
    _tls = threading.local()

    def __init__(self, handle, free):
        self.handle, self._free = handle, free

    def __del__(self):
        try:
            self._free(self.handle)
        except Exception:
            pass

    @classmethod
    def get(cls, key, make, free):
        c = getattr(cls._tls, key, None)
        if c is None:
            h = make()
            if not h:
                return None
            c = cls(h, free)
            setattr(cls._tls, key, c)
        return c.handle


class _ZBuf(C.Structure):
    _fields_ = [('p', C.c_void_p), ('size', C.c_size_t), ('pos', C.c_size_t)]


def _zstd_host_decompress(data, decompressor_context=None, size_hint=0, into=None):
    """Stream-decode a zstd frame without a content-size field (what both the reference and this library write,
    recode_writer.py:177-178) with the stock decoder: the `zstandard` package when installed (the reference's own
    dependency), else libzstd through ctypes.  Host library call, like reference recode_compressors.py:46.
    size_hint: the decoded size when the caller knows it (one output buffer, no growing).  into: uint8 array that receives
    the bytes (exactly: a stream that decodes to another length raises); the return value is then the length."""
    zs = _optional('zstandard')
    if zs is not None and into is None:
        ctx = decompressor_context if hasattr(decompressor_context, 'decompressobj') else zs.ZstdDecompressor()
        return ctx.decompressobj().decompress(bytes(data))
    L = _host_lib('zstd')
    if L is None:
        raise ImportError("For compression code 1 package zstandard (or libzstd) is required.")
    src = np.frombuffer(memoryview(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
    zds = _ThreadCtx.get('zstd', L.ZSTD_createDStream, L.ZSTD_freeDStream)
    if zds is None:
        raise MemoryError("ZSTD_createDStream")
    if into is not None:
        # the decoded size is known and the buffer is the caller's: one call, straight into it, no window buffer in between
        r = L.ZSTD_decompressDCtx(zds, into.ctypes.data, into.size, src.ctypes.data, len(data))
        if L.ZSTD_isError(r):
            raise ValueError("libzstd rejected the stream (or it decodes to more bytes than expected)")
        if r != into.size:
            raise ValueError("stream decodes to fewer bytes than expected")
        return int(r)
    out = np.empty(max(int(size_hint), 1 << 16) + 64, np.uint8)
    L.ZSTD_initDStream(zds)
    ib, got = _ZBuf(src.ctypes.data, len(data), 0), 0
    while True:
        ob = _ZBuf(out.ctypes.data + got, out.size - got, 0)
        r = L.ZSTD_decompressStream(zds, C.byref(ob), C.byref(ib))
        if L.ZSTD_isError(r):
            raise ValueError("libzstd rejected the stream")
        got += ob.pos
        if r == 0 and ib.pos == ib.size:
            break
        if ib.pos == ib.size and ob.pos < ob.size:
            raise ValueError("truncated zstd frame")
        if got == out.size:
            if into is not None:
                raise ValueError("stream decodes to more bytes than expected")
            out = np.concatenate([out, np.empty(out.size, np.uint8)])
    return out[:got].tobytes()


def _lz4_host_decompress(data, size_hint=0, into=None):
    """LZ4 frame -> bytes with the stock decoder: the `lz4` package when installed (the reference's dependency,
    recode_compressors.py:49), else liblz4's LZ4F streaming API through ctypes.  None: neither is available."""
    if _optional('lz4') is not None and into is None:
        import lz4.frame
        return lz4.frame.decompress(bytes(data))
    L = _host_lib('lz4')
    if L is None:
        return None
    def make():
        c = C.c_void_p()
        return None if L.LZ4F_isError(L.LZ4F_createDecompressionContext(C.byref(c), 100)) else c.value
    ctx = _ThreadCtx.get('lz4f', make, L.LZ4F_freeDecompressionContext)
    if ctx is None:
        return None
    L.LZ4F_resetDecompressionContext(ctx)
    src = np.frombuffer(memoryview(data), np.uint8)
    out, got, pos = (into if into is not None else np.empty(max(int(size_hint), 1 << 16) + 64, np.uint8)), 0, 0
    try:
        while pos < src.size:
            dn, sn = C.c_size_t(out.size - got), C.c_size_t(src.size - pos)
            r = L.LZ4F_decompress(ctx, out.ctypes.data + got, C.byref(dn), src.ctypes.data + pos, C.byref(sn), None)
            if L.LZ4F_isError(r):
                raise ValueError("liblz4 rejected the stream")
            got += dn.value
            pos += sn.value
            if r == 0:
                break
            if sn.value == 0 and dn.value == 0:
                raise ValueError("truncated LZ4 frame")
            if got == out.size and pos < src.size:
                if into is not None:
                    # (an exactly filled buffer: the frame's end mark and checksum may still be unread)
                    dn, sn = C.c_size_t(0), C.c_size_t(src.size - pos)
                    r = L.LZ4F_decompress(ctx, out.ctypes.data, C.byref(dn), src.ctypes.data + pos, C.byref(sn), None)
                    if L.LZ4F_isError(r) or r != 0:
                        raise ValueError("stream decodes to more bytes than expected")
                    break
                out = np.concatenate([out, np.empty(out.size, np.uint8)])
    finally:
        L.LZ4F_resetDecompressionContext(ctx)       # (an error leaves the context mid-frame)
    if into is not None:
        if got != into.size:
            raise ValueError("stream decodes to fewer bytes than expected")
        return got
    return out[:got].tobytes()


def host_stream_decoder(scheme):
    """(bytes-like, decoded size or 0, into=None or the uint8 array to fill) -> bytes (or the length) through the STOCK library: for schemes whose foreign streams the device decoders refuse
    (1 zstd, 2 LZ4) and for the host-only ones of the standard library (0 zlib, 4 bz2, 5 lzma); None when no stock decoder can be had.  Thread-safe: one decoding context per thread (_ThreadCtx)."""
    if scheme == 1:
        if _optional('zstandard') is None and _host_lib('zstd') is None:
            return None
        return lambda b, n=0, into=None: _zstd_host_decompress(b, None, n, into)
    if scheme == 2:
        if _optional('lz4') is None and _host_lib('lz4') is None:
            return None
        return lambda b, n=0, into=None: _lz4_host_decompress(b, n, into)
    if scheme in (0, 4, 5):   # host-only schemes of the standard library (zlib is the reference's own test configuration)
        fn = {0: zlib.decompress, 4: bz2.decompress, 5: lzma.decompress}[scheme]

        def dec(b, n=0, into=None):
            out = fn(bytes(b))
            if into is None:
                return out
            if len(out) != into.size:
                raise ValueError("stream decodes to %d bytes, %d expected" % (len(out), into.size))
            into[:] = np.frombuffer(out, np.uint8)
            return len(out)
        return dec
    return None


def _as_u8(data):
    return np.frombuffer(memoryview(data), dtype=np.uint8)


def device_compress(scheme, level, data):
    src = _as_u8(data)
    cap = int(_lib.lib().rc_compress_bound(int(scheme), src.size))
    dst = np.empty(cap, np.uint8)
    n = C.c_uint64(0)
    _lib.check(_lib.lib().rc_compress(int(scheme), int(level), src.ctypes.data if src.size else None, src.size,
                                      dst.ctypes.data, cap, C.byref(n)), "rc_compress")
    return dst[:n.value].tobytes()


def device_decompress(scheme, data, size_hint=0):
    src = _as_u8(data)
    n = C.c_uint64(0)
    cap = int(size_hint) if size_hint else max(4 * src.size, 1 << 16)
    for _ in range(2):
        dst = np.empty(max(cap, 1), np.uint8)
        st = _lib.lib().rc_decompress(int(scheme), src.ctypes.data, src.size, dst.ctypes.data, cap, C.byref(n))
        if st == _lib.RC_ERR_OUT_TOO_SMALL and n.value > cap:
            cap = n.value
            continue
        _lib.check(st, "rc_decompress")
        return dst[:n.value].tobytes()
    _lib.check(st, "rc_decompress")


def compress(compression_scheme, compression_level, data, compressor_context):
    s = compression_scheme
    if s not in _compression_scheme_code_map:
        raise NotImplementedError('compression scheme not implemented')
    if _on_device(s):
        return device_compress(s, compression_level, data)
    if s == 0:
        return zlib.compress(data, compression_level)
    if s == 1:
        return compressor_context.compress(data)
    if s == 3:
        return _need('snappy').compress(data)
    if s == 4:
        return bz2.compress(data, compresslevel=compression_level)
    if s == 5:
        return lzma.compress(data, preset=compression_level)
    blosc = _need('blosc')
    return blosc.compress(data, clevel=compression_level, cname=_BLOSC_CNAMES[s], shuffle=blosc.BITSHUFFLE)


def de_compress(compression_scheme, compressed_data, decompressor_context):
    s = compression_scheme
    if s not in _compression_scheme_code_map:
        raise NotImplementedError('compression scheme not implemented')
    if s in _DEVICE_DECODERS:
        return device_decompress(s, compressed_data)
    if s == 0:
        return zlib.decompress(compressed_data)
    if s == 1:
        if len(compressed_data):
            try:
                return device_decompress(1, compressed_data)
            except (NotImplementedError, ValueError):
                # UNSUPPORTED: a frame outside the device decoder's subset.  CORRUPT: the device decoder could not follow the
                # stream; whether the stream is damaged or merely foreign is the stock decoder's call (it raises on real damage).
                pass
        return _zstd_host_decompress(compressed_data, decompressor_context)
    if s == 3:
        return _need('snappy').decompress(compressed_data)
    if s == 4:
        return bz2.decompress(compressed_data)
    if s == 5:
        return lzma.decompress(compressed_data)
    return _need('blosc').decompress(compressed_data, as_bytearray=True)


def _need(module):
    m = _optional(module)
    if m is None:
        raise ImportError("For this compression scheme package " + module + " is required.")
    return m


def import_checks(header):
    """True when the package a file's compression scheme needs on the HOST is importable (device codecs need none)."""
    s = int(header['compression_scheme'])
    if s in _DEVICE_DECODERS or s in (0, 4, 5):
        return True
    if s == 1:
        import ctypes.util
        if _optional('zstandard') is None and _host_lib('zstd') is None:
            print("For compression code 1 package zstandard is required.")
            raise ImportError()
        return True
    module = {3: 'snappy'}.get(s, 'blosc')
    if _optional(module) is None:
        print("For compression code " + str(s) + " package " + _compression_scheme_code_map[s] + " is required.")
        raise ImportError()
    return True
