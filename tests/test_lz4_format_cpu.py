"""CPU: the LZ4 parses of the device encoder, restated serially (tests/lz4_parse_model.py), judged by STOCK liblz4 - every
block must decode to its input - and their effectiveness on SURVEY 8d data (what DESIGN.md quotes).  The device's bytes are
compared with this model block for block in tests/test_gpu_parity.py."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

import lz4_parse_model as model
from pyrecode_amd import synth


@pytest.fixture(scope="module")
def lz4():
    name = ctypes.util.find_library("lz4")
    if not name:
        pytest.skip("no system liblz4")
    L = C.CDLL(name)
    L.LZ4_decompress_safe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    return L


def _decode(lz4, payload, n):
    src = np.frombuffer(payload, np.uint8)
    dst = np.empty(max(n, 1), np.uint8)
    r = lz4.LZ4_decompress_safe(src.ctypes.data, dst.ctypes.data, len(payload), n)
    assert r == n, r
    return dst[:n].tobytes()


def _blocks():
    rng = np.random.default_rng(3)
    out = []
    for p in (0.0, 0.002, 0.01, 0.015, 0.02, 0.025, 0.03, 0.05, 0.12, 0.3):   # sparse bitmaps up to dense ones (one event per lane up to 62
                                                                              # events, two up to 126, the run parser beyond)
        for _ in range(40):
            bits = rng.random(4096) < p
            out.append(np.packbits(bits, bitorder="little").tobytes())
    for n in (1, 4, 11, 12, 13, 16, 17, 29, 100, 511):         # short last blocks of a stream
        blk = np.zeros(n, np.uint8)
        out.append(blk.tobytes())
        blk[rng.integers(0, n, max(n // 9, 1))] = 1 << rng.integers(0, 8)
        out.append(blk.tobytes())
    one = np.zeros(512, np.uint8)
    for pos in (0, 1, 255, 499, 500, 505, 506, 507, 511):      # a single event, also inside the last 12 / 5 bytes
        b = one.copy()
        b[pos] = 0x10
        out.append(b.tobytes())
    rep = np.zeros(512, np.uint8)                               # the same unit again and again: every match reaches its source's end
    rep[::7] = 4
    out.append(rep.tobytes())
    rep2 = np.zeros(512, np.uint8)                              # growing gaps, one value: leads and trails of every length
    q = 0
    for g in range(1, 40):
        q += g
        if q < 512:
            rep2[q] = 0x80
    out.append(rep2.tobytes())
    for period, val in ((5, 0x20), (6, 0x01), (7, 0x80)):        # 73 - 103 events of one value: two per lane, every one with a unit copy
        r = np.zeros(512, np.uint8)
        r[::period] = val
        out.append(r.tobytes())
    r = np.zeros(512, np.uint8)                                  # the same with the eight values in turn and two gap lengths
    r[::5] = 1 << (np.arange(103) % 8)
    r[3::35] = 0x40
    out.append(r.tobytes())
    dark = synth.dark_frame(11, 512 * 512)                       # detector-like clusters (bytes with several set bits between the single-bit ones)
    cl = np.packbits(synth.frames_clustered(11, 0, 1, 512, 512, 11000, dark)[0] > dark, bitorder="little")
    out += [cl[i:i + 512].tobytes() for i in range(0, cl.size, 512)]
    multi = np.zeros(512, np.uint8)                             # bytes with two set bits have no class: literals
    multi[::9] = 0x81
    out.append(multi.tobytes())
    out.append(bytes(512))
    out.append(bytes([0xFF]) * 512)
    return out


@pytest.mark.parametrize("level", [0, 1])
def test_model_blocks_decode_with_stock_liblz4(lz4, level):
    for blk in _blocks():
        m = model.parse_events(blk) if level else None
        if m is None:
            m = model.parse_runs(blk)
        pos = 0
        for s, l, off in m:                                      # position order, no overlap, LZ4's block-end rules
            assert s >= pos and l >= 4 and 1 <= off <= s and s + 12 <= len(blk) and s + l <= len(blk) - 5
            pos = s + l
        assert _decode(lz4, model.emit(blk, m), len(blk)) == blk


def test_event_parse_ratio_on_survey_data(lz4):
    """4096 x 4096 at 1 %: the binary map in independent 512-byte blocks, 4-byte block words included."""
    N = 4096 * 4096
    dark = synth.dark_frame(20261003, N)
    bm = np.packbits(synth.frames(20261003, 0, 1, N, 10000, dark)[0] > dark, bitorder="little")
    tot = {0: 0, 1: 0}
    picked = range(0, bm.size // 512, 13)
    for i in picked:
        blk = bm[i * 512:(i + 1) * 512].tobytes()
        for level in (0, 1):
            w, payload = model.encode_block(blk, level)
            if not w & 0x80000000:
                assert _decode(lz4, payload, 512) == blk
            tot[level] += 4 + len(payload)
    raw = len(picked) * 512
    assert 0.36 < tot[0] / raw < 0.39          # the run parser (round 1 / 2: 0.375)
    assert tot[1] / raw < 0.30                 # the event parser


def test_event_parse_ratio_on_detector_like_maps(lz4):
    """Dense maps (the reference notebook's acquisition: 4.3 % set pixels in clusters; 2 % Bernoulli): most blocks hold 63 - 126 events, the
    two-events-per-lane form of the parser.  Stock liblz4 on the same maps: 0.47 / 0.43 with 64 KiB blocks, 0.555 / 0.49 on independent
    512-byte blocks; the run parser alone: 0.62 / 0.59."""
    nx = ny = 1024
    dark = synth.dark_frame(7, nx * ny)
    for frames, bound, runs_bound in ((synth.frames_clustered(7, 0, 1, nx, ny, 11000, dark), 0.54, 0.60), (synth.frames(7, 0, 1, nx * ny, 20000, dark), 0.45, 0.57)):
        bm = np.packbits(frames[0] > dark, bitorder="little")
        tot = {0: 0, 1: 0}
        two = 0
        for i in range(bm.size // 512):
            blk = bm[i * 512:(i + 1) * 512].tobytes()
            two += model.EV_MAX < np.count_nonzero(bm[i * 512:(i + 1) * 512]) <= model.EV_MAX2
            for level in (0, 1):
                w, payload = model.encode_block(blk, level)
                if not w & 0x80000000:
                    assert _decode(lz4, payload, 512) == blk
                tot[level] += 4 + len(payload)
        assert two > 0.9 * (bm.size // 512)
        assert tot[1] / bm.size < bound and tot[0] / bm.size > runs_bound
