"""Serial restatement (test infrastructure) of the device DEFLATE block encoder (pyrecode_amd/csrc/rc_deflate_block.h): the matches
of the LZ4 parsers (tests/lz4_parse_model.py: the same parse serves both formats) written as ONE fixed-Huffman block (RFC 1951,
BTYPE 01) per 512-byte tile, closed by an empty stored block so that the next tile's block starts on a byte boundary (zlib's
Z_SYNC_FLUSH marker); the frame's last tile carries BFINAL and is padded to the byte instead.  A tile that would not shrink is a
stored block.  Used two ways: on the CPU stdlib zlib must expand what the model emits (the stream is format-conformant); on the GPU
the device's bytes must equal the model's, tile for tile."""
import zlib

import numpy as np

import lz4_parse_model as lz4m

NZ_STORE = lz4m.NZ_STORE
CHUNK = 1 << 15     # stored blocks of the residual stream (rc_record.h::frame_fmt)


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):          # LSB-first (extra bits, headers)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def put_code(self, code, nbits):      # Huffman codes go in MSB-first
        self.put(int(format(code, "0%db" % nbits)[::-1], 2), nbits)

    def bits(self):
        return 8 * len(self.out) + self.n

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def lit_code(b):
    return (0x30 + b, 8) if b < 144 else (0x190 + b - 144, 9)


def len_symbol(length):
    """(symbol, extra bits, extra value) of a match length 3 .. 258"""
    l = length - 3
    if l == 255:
        return 285, 0, 0
    if l < 8:
        return 257 + l, 0, 0
    e = l.bit_length() - 3
    return 257 + 4 * (e + 1) + ((l >> e) & 3), e, l & ((1 << e) - 1)


def dist_symbol(dist):
    d = dist - 1
    if d < 4:
        return d, 0, 0
    hb = d.bit_length() - 1
    e = hb - 1
    return 2 * hb + ((d >> e) & 1), e, d & ((1 << e) - 1)


def sym_code(sym):
    return (sym - 256, 7) if sym < 280 else (0xC0 + sym - 280, 8)


def split_match(length):
    """a match of up to 512 bytes as one or two DEFLATE matches (3 .. 258 each)"""
    if length <= 258:
        return [length]
    first = 258 if length - 258 >= 3 else length - 3
    return [first, length - first]


def emit_fixed(block, matches, last):
    w = BitWriter()
    w.put((1 if last else 0) | (1 << 1), 3)
    pos = 0
    for s, l, off in matches:
        for b in block[pos:s]:
            w.put_code(*lit_code(b))
        for part in split_match(l):
            sym, e, ev = len_symbol(part)
            w.put_code(*sym_code(sym))
            w.put(ev, e)
            ds, de, dv = dist_symbol(off)
            w.put_code(ds, 5)
            w.put(dv, de)
        pos = s + l
    for b in block[pos:]:
        w.put_code(*lit_code(b))
    w.put_code(0, 7)                     # end of block
    if not last:
        w.put(0, 3)                      # an empty stored block ...
        w.align()
        w.out += b"\x00\x00\xff\xff"     # ... LEN 0, NLEN: the next tile starts on a byte
    else:
        w.align()
    return bytes(w.out)


def stored(block, last):
    n = len(block)
    return bytes([1 if last else 0, n & 255, n >> 8, ~n & 255, (~n >> 8) & 255]) + bytes(block)


def encode_tile(block, last=False, level=1):
    """the bytes one tile contributes to the frame's deflate stream"""
    block = bytes(block)
    if level and int(np.count_nonzero(np.frombuffer(block, np.uint8))) > NZ_STORE:
        return stored(block, last)
    m = lz4m.parse_events(block) if level else None
    if m is None:
        m = lz4m.parse_runs(block)
    enc = emit_fixed(block, m, last)
    return enc if len(enc) < len(block) + 5 else stored(block, last)


def adler32(data):
    return zlib.adler32(data) & 0xFFFFFFFF


def bitmap_stream(bitmap, level=1):
    """the zlib stream of a frame's packed binary map: header, a block (pair) per 512-byte tile, Adler-32"""
    bitmap = bytes(bitmap)
    nt = max((len(bitmap) + 511) // 512, 1)
    out = bytearray(b"\x78\x01")
    for t in range(nt):
        out += encode_tile(bitmap[512 * t:512 * (t + 1)], t + 1 == nt, level)
    return bytes(out) + adler32(bitmap).to_bytes(4, "big")


def stored_stream(data):
    """the zlib stream of the packed residuals: stored blocks of 32 KiB"""
    data = bytes(data)
    nch = max((len(data) + CHUNK - 1) // CHUNK, 1)
    out = bytearray(b"\x78\x01")
    for k in range(nch):
        out += stored(data[k * CHUNK:(k + 1) * CHUNK], k + 1 == nch)
    return bytes(out) + adler32(data).to_bytes(4, "big")
