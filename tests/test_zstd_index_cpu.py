"""CPU, under AddressSanitizer: the batched reader's host-side frame walker (rc_zstd_dec.h::zd_index_frame) on frames this library's
encoder model writes, on frames the STOCK libzstd writes, and on truncations / byte flips of both.  It may answer OK, FOREIGN or
CORRUPT - never read outside the stream, never hand out a block entry that points outside it."""
import ctypes as C
import ctypes.util
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("zdidx")
    exe = d / "zd_index_harness"
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                           os.path.join(REPO, "tests", "native", "zd_index_harness.cpp")])
    so = d / "libzstd_check.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", str(so), os.path.join(REPO, "tests", "native", "zstd_host_check.cpp")])
    enc = C.CDLL(str(so))
    enc.zstd_check_encode_frame.restype = C.c_int64
    enc.zstd_check_encode_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    return exe, enc, d


def _own_frame(enc, data):
    src = np.frombuffer(data, np.uint8)
    dst = np.empty(len(data) + len(data) // 32 + 256, np.uint8)
    n = enc.zstd_check_encode_frame(src.ctypes.data, src.size, dst.ctypes.data, dst.size, 1)
    assert n > 0
    return dst[:n].tobytes()


def _stock_frame(data, level):
    name = ctypes.util.find_library("zstd")
    if not name:
        return None
    z = C.CDLL(name)
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    dst = C.create_string_buffer(len(data) + len(data) // 8 + 1024)
    n = z.ZSTD_compress(dst, len(dst), data, len(data), level)
    return dst.raw[:n]


def _run(harness, cases):
    exe, _, d = harness
    path = d / "cases.bin"
    with open(path, "wb") as f:
        for frame, expect, total in cases:
            f.write(struct.pack("<IIQ", len(frame), expect, total) + frame)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([str(exe), str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stderr.decode()[-3000:], p.stdout.decode()[-300:])
    rows = [tuple(int(v) for v in line.split()) for line in p.stdout.decode().splitlines()]
    assert len(rows) == len(cases)
    return rows


def test_walker_on_own_frames_stock_frames_and_damage(harness):
    _, enc, _ = harness
    rng = np.random.default_rng(11)
    cases, kinds = [], []
    UNKNOWN = 0xFFFFFFFFFFFFFFFF
    for nbytes, dens in ((512, 0.01), (513, 0.05), (5000, 0.02), (32768, 0.0003), (32768, 0.01), (70001, 0.002), (1, 0.5), (4096, 0.0)):
        data = np.packbits(rng.random(nbytes * 8) < dens, bitorder="little").tobytes()
        own = _own_frame(enc, data)
        cases.append((own, 512, nbytes)); kinds.append("own")
        cases.append((own, 512, UNKNOWN)); kinds.append("own-open")
        for level in (1, 3, 19):
            st = _stock_frame(data, level)
            if st is not None:
                cases.append((st, 512, nbytes)); kinds.append("stock")
                cases.append((st, 512, UNKNOWN)); kinds.append("stock-open")
    base = list(zip(cases, kinds))
    for (frame, expect, total), kind in base:            # damage: every truncation of the first 200 bytes, then sparse ones; byte flips
        cuts = list(range(0, min(len(frame), 200))) + list(range(200, len(frame), max(1, len(frame) // 40)))
        for c in cuts:
            cases.append((frame[:c], expect, total)); kinds.append("cut")
        for _ in range(60):
            b = bytearray(frame)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            cases.append((bytes(b), expect, total)); kinds.append("flip")
        cases.append((frame + frame, expect, total)); kinds.append("two frames")
        cases.append((frame + b"\x50\x2a\x4d\x18\x04\x00\x00\x00abcd", expect, total)); kinds.append("skippable behind")
    rows = _run(harness, cases)
    for (st, nblk, regen), (frame, expect, total), kind in zip(rows, cases, kinds):
        assert st in (0, -1, -2), (kind, st)
        if kind == "own":
            assert st == 0 and regen == total and nblk == (total + 511) // 512, (kind, st, nblk, regen, total)
        if kind == "own-open":
            assert st == 0 and nblk >= 1
        if kind in ("two frames", "skippable behind"):
            assert st in (-2, -1)
        if kind == "cut" and len(frame) < 6:
            assert st != 0
