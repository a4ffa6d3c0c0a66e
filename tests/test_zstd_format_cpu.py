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
