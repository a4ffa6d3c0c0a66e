"""CPU: the stock-library decoders the reader falls back to for streams a FOREIGN encoder wrote (pyrecode_amd/recode_compressors.py:
host_stream_decoder - the same library calls the reference makes, recode_compressors.py:46-49): frames of the system's libzstd / liblz4
decode through the ctypes bindings, into a caller's buffer of the known size as well, and damaged or mis-sized streams raise."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

from pyrecode_amd import recode_compressors as rcomp


def _stock_encoders():
    enc = {}
    name = ctypes.util.find_library("zstd")
    if name:
        z = C.CDLL(name)
        z.ZSTD_compress.restype = C.c_size_t
        z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]

        def zenc(b, level=1):
            dst = C.create_string_buffer(len(b) + len(b) // 8 + 1024)
            n = z.ZSTD_compress(dst, len(dst), b, len(b), level)
            return dst.raw[:n]
        enc[1] = zenc
    name = ctypes.util.find_library("lz4")
    if name:
        lz = C.CDLL(name)
        lz.LZ4F_compressFrameBound.restype = C.c_size_t
        lz.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
        lz.LZ4F_compressFrame.restype = C.c_size_t
        lz.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]

        def lenc(b, level=1):
            dst = C.create_string_buffer(lz.LZ4F_compressFrameBound(len(b), None) + 64)
            n = lz.LZ4F_compressFrame(dst, len(dst), b, len(b), None)
            return dst.raw[:n]
        enc[2] = lenc
    return enc


@pytest.mark.parametrize("scheme", [1, 2])
def test_stock_frames_decode_through_the_ctypes_bindings(scheme):
    enc = _stock_encoders().get(scheme)
    if enc is None:
        pytest.skip("no system library for scheme %d" % scheme)
    dec = rcomp.host_stream_decoder(scheme)
    assert dec is not None
    rng = np.random.default_rng(scheme)
    payloads = [b"", b"\x07", bytes(70000),                                                         # empty, one byte, a long zero run
                np.packbits(rng.random(300000 * 8) < 0.01).tobytes(),                               # a sparse binary map (several 64 KiB blocks)
                rng.integers(0, 2048, 50000).astype("<u2").tobytes(),                               # residuals: incompressible low bytes
                rng.integers(0, 256, 200000).astype(np.uint8).tobytes()]                            # noise: stored blocks
    for data in payloads:
        comp = enc(data)
        assert dec(comp) == data
        assert dec(comp, len(data)) == data
        if data:
            out = np.full(len(data), 0xAA, np.uint8)
            assert dec(comp, len(data), out) == len(data) and out.tobytes() == data
            with pytest.raises(ValueError):                                                          # the caller's size is a contract
                dec(comp, len(data) - 1, np.empty(len(data) - 1, np.uint8))
            with pytest.raises(ValueError):
                dec(comp, len(data) + 1, np.empty(len(data) + 1, np.uint8))
            with pytest.raises(ValueError):                                                          # truncated stream
                dec(comp[:len(comp) // 2], len(data), np.empty(len(data), np.uint8))
    threads = 8                                                                                      # every call builds its own context
    from concurrent.futures import ThreadPoolExecutor
    comp = enc(payloads[3])
    with ThreadPoolExecutor(threads) as pool:
        assert all(r == payloads[3] for r in pool.map(lambda _: dec(comp, len(payloads[3])), range(32)))


def test_standard_library_schemes_and_unknown_ones():
    import bz2, lzma, zlib
    data = np.packbits(np.random.default_rng(5).random(40000 * 8) < 0.02).tobytes()
    for scheme, enc in ((0, lambda b: zlib.compress(b, 1)), (4, bz2.compress), (5, lzma.compress)):
        dec = rcomp.host_stream_decoder(scheme)
        out = np.zeros(len(data), np.uint8)
        assert dec(enc(data)) == data and dec(enc(data), len(data), out) == len(data) and out.tobytes() == data
        with pytest.raises(ValueError):
            dec(enc(data), len(data) + 3, np.zeros(len(data) + 3, np.uint8))
    assert rcomp.host_stream_decoder(8) is None and rcomp.host_stream_decoder(3) is None


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_native_batch_decoder_makes_the_stock_librarys_calls(scheme):
    """rc_host_decode_streams (include/recode_hip.h): n streams of a stock encoder at once on the library's worker threads, each to its
    exact place and size; a damaged stream, a stream of another length and trailing bytes are RC_ERR_CORRUPT.  No GPU involved."""
    import zlib
    from pyrecode_amd import _lib
    L = _lib.lib()
    enc = (lambda b: zlib.compress(b, 1)) if scheme == 0 else _stock_encoders().get(scheme)
    if enc is None or not L.rc_host_decoder_available(scheme):
        pytest.skip("no system library for scheme %d" % scheme)
    rng = np.random.default_rng(40 + scheme)
    payloads = [np.packbits(rng.random(n * 8) < p).tobytes() for n, p in ((300000, 0.01), (70000, 0.002), (513, 0.3), (1, 0.5), (150000, 0.05))]
    payloads += [rng.integers(1, 2048, 40000).astype("<u2").tobytes(), bytes(100000), b""]
    payloads = payloads * 5                                            # 40 streams: more than the worker threads
    comp = [enc(p) for p in payloads]
    src = np.frombuffer(b"".join(comp), np.uint8)
    spans, so, do = [], 0, 7                                           # (an unaligned destination)
    for c, p in zip(comp, payloads):
        spans.append((so, len(c), do, len(p)))
        so += len(c)
        do += len(p) + 3                                               # 3 guard bytes behind every stream
    table = np.array(spans, np.uint64)
    for threads in (0, 1, 3):
        dst = np.full(do, 0xA5, np.uint8)
        assert L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(table), len(spans), threads) == _lib.RC_OK
        for (a, n, at, want), p in zip(spans, payloads):
            assert dst[at:at + want].tobytes() == p
            assert (dst[at + want:at + want + 3] == 0xA5).all()
    assert L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(table), 0, 0) == _lib.RC_OK
    # refusals
    dst = np.zeros(do, np.uint8)
    bad = table.copy(); bad[0, 3] -= 1                                 # decodes to MORE than expected
    assert L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(bad), len(spans), 0) == _lib.RC_ERR_CORRUPT
    assert "stream 0" in _lib.last_error()
    bad = table.copy(); bad[4, 3] += 1                                 # ... to fewer
    assert L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(bad), len(spans), 0) == _lib.RC_ERR_CORRUPT
    bad = table.copy(); bad[1, 1] -= 5                                 # truncated
    assert L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(bad), len(spans), 0) == _lib.RC_ERR_CORRUPT
    if scheme != 0:                                                    # (zlib.decompress itself ignores nothing either, but uncompress() stops at the end mark)
        bad = table.copy(); bad[2, 1] += 4                             # bytes behind the frame
        assert L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(bad), len(spans), 0) == _lib.RC_ERR_CORRUPT
    flipped = src.copy(); flipped[int(table[0, 0]) + int(table[0, 1]) // 2] ^= 0x10
    st = L.rc_host_decode_streams(scheme, _lib.ptr(flipped), _lib.ptr(dst), _lib.ptr(table), len(spans), 0)
    assert st in (_lib.RC_OK, _lib.RC_ERR_CORRUPT)                     # (a flipped literal still decodes; it must not crash or overrun)
    assert L.rc_host_decode_streams(7, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(table), len(spans), 0) == _lib.RC_ERR_UNSUPPORTED


def test_split_triplets_equals_the_three_numpy_conversions():
    """rc_split_triplets: (row, col, value) uint64 rows -> int32 rows, int32 columns, values of 1 / 2 / 4 / 8 bytes, one pass (what
    ReCoDeReader._make_coo_frame hands to the COO matrix, reference recode_reader.py:466-469); n = 0 and a bad width."""
    from pyrecode_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for n in (0, 1, 1000, 70001):
        trip = np.empty((n, 3), np.uint64)
        trip[:, 0], trip[:, 1], trip[:, 2] = rng.integers(0, 8184, n), rng.integers(0, 11520, n), rng.integers(0, 1 << 16, n)
        for width, dt in ((1, np.uint8), (2, np.uint16), (4, np.uint32), (8, np.uint64), (2, np.int16), (4, np.int32)):
            row, col, val = np.full(n + 1, -5, np.int32), np.full(n + 1, -5, np.int32), np.full(n + 1, 77, dt)
            assert L.rc_split_triplets(_lib.ptr(trip) if n else None, n, _lib.ptr(row), _lib.ptr(col), _lib.ptr(val), width) == _lib.RC_OK
            assert np.array_equal(row[:n], trip[:, 0].astype(np.int32)) and np.array_equal(col[:n], trip[:, 1].astype(np.int32))
            assert np.array_equal(val[:n], trip[:, 2].astype(dt)) and row[n] == -5 and col[n] == -5 and val[n] == 77
    assert L.rc_split_triplets(_lib.ptr(trip), n, _lib.ptr(row), _lib.ptr(col), _lib.ptr(val), 3) == _lib.RC_ERR_BAD_ARG
