"""CPU: the device DEFLATE encoder restated serially (tests/deflate_block_model.py), judged by STDLIB zlib - the decoder the reference's
reader calls (pyrecode/recode_compressors.py:43) - on the block set of the LZ4 format tests, on short tails, on whole frames of SURVEY 8d
data, and its size on those (what DESIGN.md quotes).  The device's bytes are compared with this model tile for tile in
tests/test_gpu_deflate.py."""
import zlib

import numpy as np
import pytest

import deflate_block_model as model
from pyrecode_amd import synth
from test_lz4_format_cpu import _blocks


def _inflate_raw(data):
    d = zlib.decompressobj(-15)
    out = d.decompress(data) + d.flush()
    return out, d.unused_data, d.eof


def test_symbol_tables_cover_every_length_and_distance():
    """every match length 3..258 and distance 1..512 lands in the symbol RFC 1951 3.2.5 assigns it"""
    base_len = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    extra_len = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
    for L in range(3, 259):
        sym, e, v = model.len_symbol(L)
        i = sym - 257
        assert e == extra_len[i] and base_len[i] + v == L and v < (1 << e), L
    base_d = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513]
    for D in range(1, 513):
        sym, e, v = model.dist_symbol(D)
        assert e == max(sym // 2 - 1, 0) and base_d[sym] + v == D and v < (1 << e), D
    for L in range(4, 513):
        parts = model.split_match(L)
        assert sum(parts) == L and all(3 <= p <= 258 for p in parts) and len(parts) <= 2


@pytest.mark.parametrize("last", [False, True])
def test_model_tiles_inflate_with_stdlib_zlib(last):
    for blk in _blocks():
        enc = model.encode_tile(blk, last)
        assert len(enc) <= len(blk) + 5
        if last:
            out, unused, eof = _inflate_raw(enc)
            assert out == blk and eof and unused == b""
        else:   # a tile in the middle of a stream: the next block starts on a byte; close the stream with an empty final block
            out, unused, eof = _inflate_raw(enc + b"\x01\x00\x00\xff\xff")
            assert out == blk and eof and unused == b""
            assert enc[-4:] == b"\x00\x00\xff\xff" or enc[0] == 0     # the sync marker, or a stored block


def test_model_streams_decompress_with_stdlib_zlib():
    rng = np.random.default_rng(5)
    for n, p in ((1, 0.5), (7, 0.1), (511, 0.02), (512, 0.0), (513, 0.01), (1024, 1.0), (5000, 0.03), (40000, 0.01), (70001, 0.3)):
        bits = rng.random(8 * n) < p
        bm = np.packbits(bits, bitorder="little").tobytes()
        assert zlib.decompress(model.bitmap_stream(bm)) == bm
    for data in (b"", b"\x01", bytes(32767), bytes(32768), bytes(range(256)) * 129, rng.integers(0, 256, 100000, np.uint8).tobytes()):
        s = model.stored_stream(data)
        assert zlib.decompress(s) == data
        assert len(s) == 2 + len(data) + 5 * max(-(-len(data) // model.CHUNK), 1) + 4


def test_ratio_on_survey_data():
    """4096 x 4096 at 1 % (a sixteenth of the frame): fixed-Huffman blocks per tile with the sync marker against stock zlib."""
    N = 1024 * 4096
    dark = synth.dark_frame(20261003, N)
    bm = np.packbits(synth.frames(20261003, 0, 1, N, 10000, dark)[0] > dark, bitorder="little").tobytes()
    s = model.bitmap_stream(bm)
    assert zlib.decompress(s) == bm
    ratio, stock = len(s) / len(bm), len(zlib.compress(bm, 1)) / len(bm)
    assert 0.20 < ratio < 0.225 and 0.16 < stock < 0.18, (ratio, stock)    # (the LZ4 device stream of the same map: 0.281)
