"""CPU: the zstd block encoder shared by the HIP kernel (pyrecode_amd/csrc/rc_zstd_block.h), compiled for the host, must
produce frames the STOCK libzstd decoder accepts and expands to the bit-exact input."""
import ctypes as C
import ctypes.util
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def enc(tmp_path_factory):
    so = tmp_path_factory.mktemp("zstdchk") / "libzstd_check.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", str(so), os.path.join(REPO, "tests", "native", "zstd_host_check.cpp")])
    L = C.CDLL(str(so))
    L.zstd_check_encode_frame.restype = C.c_int64
    L.zstd_check_encode_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    return L


@pytest.fixture(scope="module")
def zstd():
    name = ctypes.util.find_library("zstd")
    if not name:
        pytest.skip("libzstd not installed")
    L = C.CDLL(name)
    L.ZSTD_decompress.restype = C.c_size_t
    L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_isError.argtypes = [C.c_size_t]
    L.ZSTD_getErrorName.restype = C.c_char_p
    L.ZSTD_getErrorName.argtypes = [C.c_size_t]
    return L


def _roundtrip(enc, zstd, data):
    src = np.frombuffer(data, np.uint8) if len(data) else np.zeros(0, np.uint8)
    dst = np.empty(len(data) + len(data) // 64 + 64, np.uint8)
    sizes = []
    for streaming in (0, 1):  # serial restatement, then the token form the HIP kernels run (same fse_chain code)
        n = enc.zstd_check_encode_frame(src.ctypes.data if src.size else None, src.size, dst.ctypes.data, dst.size, streaming)
        assert n > 0
        out = np.empty(len(data) + 16, np.uint8)
        r = zstd.ZSTD_decompress(out.ctypes.data, out.size, dst.ctypes.data, n)
        assert not zstd.ZSTD_isError(r), zstd.ZSTD_getErrorName(r)
        assert r == len(data) and out[:r].tobytes() == data
        sizes.append((n, dst[:n].tobytes()))
    # same bytes, except that the token form decides "store raw" from an upper bound of the bitstream length (before the
    # bitstream exists): blocks within a few bytes of not shrinking may be stored raw there and compressed by the serial form
    if sizes[0] != sizes[1]:
        assert sizes[0][0] <= sizes[1][0] <= sizes[0][0] * 1.02 + 8
    return sizes[1][0]


@pytest.mark.parametrize("density", [0.0, 0.001, 0.01, 0.077, 0.3, 0.7, 1.0])
def test_sparse_bitmaps_roundtrip(enc, zstd, density):
    rng = np.random.default_rng(int(density * 1000))
    for n in (1, 3, 4, 5, 511, 512, 513, 2048, 100000):
        data = np.where(rng.random(n) < density, rng.integers(1, 256, n), 0).astype(np.uint8).tobytes()
        c = _roundtrip(enc, zstd, data)
        if density <= 0.01 and n >= 2048:
            assert c < 0.2 * n


def test_run_length_edges(enc, zstd):
    # runs of every length around the code-table boundaries (LL codes 16.., ML codes 32.., extra bits), at both block ends
    for run in list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 130, 131, 255, 256, 257, 258, 259, 260, 400, 509, 510, 511]:
        for lead in (0, 1, 2, 15, 16, 17, 31, 32, 33):
            data = bytes([7] * lead + [0] * run + [9])
            if len(data) <= 512:
                data = data + bytes([1] * (512 - len(data)))
            _roundtrip(enc, zstd, data)
    _roundtrip(enc, zstd, b"")
    _roundtrip(enc, zstd, bytes(512))
    _roundtrip(enc, zstd, bytes(512) + b"\x01")
    _roundtrip(enc, zstd, b"\x01" + bytes(511))
    _roundtrip(enc, zstd, bytes(511) + b"\x01")
    # many short sequences in one block (>= 128 sequences needs the 2-byte count form): 0001 pattern -> runs of 3 don't match; use 00001
    _roundtrip(enc, zstd, (b"\x00\x00\x00\x00\x05" * 110)[:512])
    _roundtrip(enc, zstd, (b"\x00\x00\x00\x00" + b"\x05") * 102 + b"\x00\x00")


# ---- modelled encoder: fitted Huffman / FSE tables, defined once per frame (rc_zstd_model.h) -------------------------------
@pytest.fixture(scope="module")
def zm(enc):
    enc.zm_model_bytes.restype = C.c_uint64
    enc.zm_check_build.restype = C.c_uint32
    enc.zm_check_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    for f in (enc.zm_check_encode_bitmap_frame,):
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    enc.zm_check_encode_pix_frame.restype = C.c_int64
    enc.zm_check_encode_pix_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
    return enc


def _sparse_bitmap(rng, nbytes, density):
    bits = rng.random(nbytes * 8) < density
    return np.packbits(bits, bitorder="little")


def _model(zm, bitmap, pix):
    m = np.zeros(int(zm.zm_model_bytes()), np.uint8)
    valid = zm.zm_check_build(bitmap.ctypes.data if bitmap.size else None, bitmap.size, pix.ctypes.data if pix.size else None, pix.size,
                              m.ctypes.data)
    return m, valid


def _decode(zstd, buf, n, want_len):
    out = np.empty(want_len + 16, np.uint8)
    r = zstd.ZSTD_decompress(out.ctypes.data, out.size, buf.ctypes.data, n)
    assert not zstd.ZSTD_isError(r), zstd.ZSTD_getErrorName(r)
    return out[:r]


@pytest.mark.parametrize("density", [0.0005, 0.01, 0.05, 0.3])
def test_modelled_bitmap_frames(zm, zstd, density):
    """Blocks with treeless literals / Repeat_Mode tables + the descriptions inserted into the first block that needs them:
    stock libzstd must accept the frame and return the input; at 1 % the ratio must beat libzstd level 1 (0.143)."""
    rng = np.random.default_rng(int(density * 1e4))
    sample = _sparse_bitmap(rng, 64 * 512, density)
    pix = rng.integers(1, 2048, 4096).astype(np.uint16).view(np.uint8)
    m, valid = _model(zm, sample, pix)
    # sparse maps keep the sequences (zero runs as matches); dense ones (5 %, 30 %) come out smaller with every byte a Huffman-coded
    # literal and no sequences at all: bit 3 (ZM_LITS_ONLY), and then no sequence tables (bit 2)
    assert valid == (7 if density <= 0.01 else 11), valid
    for n in (1, 5, 511, 512, 513, 4096, 200000):
        for d in (density, 0.0, min(1.0, density * 8), 0.6):   # data like the sample, and data the model was not fitted to
            data = _sparse_bitmap(rng, n, d)[:n]
            if n >= 2048 and d == density:
                data[:700] = 0   # leading all-zero (RLE) blocks: the definitions then travel in a later block
            dst = np.empty(n + n // 32 + 1024, np.uint8)
            k = zm.zm_check_encode_bitmap_frame(data.ctypes.data, n, dst.ctypes.data, dst.size, m.ctypes.data)
            assert k > 0, k
            assert np.array_equal(_decode(zstd, dst, k, n), data)
            if n == 200000 and d == density == 0.01:
                assert k / n < 0.14, k / n
            if n == 200000 and d == density == 0.05:
                # the entropy of a 5 % Bernoulli map is 0.286 of raw; the sequences form wrote 0.355 (BASELINE cfg 5's records, round 3)
                assert k / n < 0.34, k / n


def test_modelled_bitmap_frame_with_untrained_model(zm, zstd):
    """An empty sample still yields a usable (if useless) model: every symbol stays encodable."""
    rng = np.random.default_rng(3)
    m, valid = _model(zm, np.zeros(0, np.uint8), np.zeros(0, np.uint8))
    assert valid == 7
    data = _sparse_bitmap(rng, 10000, 0.02)
    dst = np.empty(20000, np.uint8)
    k = zm.zm_check_encode_bitmap_frame(data.ctypes.data, data.size, dst.ctypes.data, dst.size, m.ctypes.data)
    assert k > 0 and np.array_equal(_decode(zstd, dst, k, data.size), data)


@pytest.mark.parametrize("kind", ["uniform11", "exp", "const", "random"])
def test_modelled_pixel_stream_frames(zm, zstd, kind):
    rng = np.random.default_rng(11)
    gen = {"uniform11": lambda k: rng.integers(1, 2048, k).astype(np.uint16).view(np.uint8),
           "exp": lambda k: np.minimum(rng.exponential(40, k), 4000).astype(np.uint16).view(np.uint8),
           "const": lambda k: np.full(k, 7, np.uint16).view(np.uint8),
           "random": lambda k: rng.integers(0, 256, 2 * k).astype(np.uint8)}[kind]
    m, valid = _model(zm, _sparse_bitmap(rng, 8192, 0.01), gen(20000))
    assert valid & 2
    for k in (0, 1, 3, 503, 504, 505, 100000):
        data = np.ascontiguousarray(gen(k))
        dst = np.empty(data.size + data.size // 64 + 1024, np.uint8)
        c = zm.zm_check_encode_pix_frame(data.ctypes.data if data.size else None, data.size, dst.ctypes.data, dst.size, m.ctypes.data, 1008)
        assert c > 0
        assert np.array_equal(_decode(zstd, dst, c, data.size), data)
        if k == 100000 and kind == "uniform11":
            assert c / data.size < 0.82   # byte-wise Huffman bound of this distribution: 0.80 (stock libzstd: 0.8008)
        if k == 100000 and kind == "random":
            assert c / data.size < 1.01   # incompressible input falls back to Raw blocks
