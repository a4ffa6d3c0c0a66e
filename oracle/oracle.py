"""ORACLE - test infrastructure, NOT the product.

ctypes binding of oracle/librecode_oracle.so (the plain-C CPU restatement, recode_oracle.c) plus the
thin numpy/stdlib glue that turns its per-stage outputs into the reference's record and file bytes.
Each function cites the reference file:line (relative to /root/reference) it restates.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Never pyrecode_amd/.
Pinned by tests/test_oracle_golden.py against fixtures captured from the reference itself.
"""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement (gcc only; no GPU, no reference needed)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librecode_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u8p, u16p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint64)
        L.orc_threshold.argtypes = [u16p, C.c_int64, C.c_uint64, u16p]
        L.orc_threshold.restype = None
        L.orc_binarize_l1.argtypes = [u16p, u16p, C.c_uint64, u8p, u16p]
        L.orc_binarize_l1.restype = C.c_uint64
        L.orc_pack_binary_frame.argtypes = [u8p, C.c_uint64, u8p]
        L.orc_pack_binary_frame.restype = None
        L.orc_bit_pack.argtypes = [u16p, C.c_uint64, C.c_uint, u8p]
        L.orc_bit_pack.restype = C.c_uint64
        L.orc_bit_unpack.argtypes = [u8p, C.c_uint64, C.c_uint, u64p]
        L.orc_bit_unpack.restype = None
        L.orc_unpack_frame_sparse.argtypes = [C.c_uint32, C.c_uint32, C.c_uint, u8p, u8p, u64p, C.c_uint]
        L.orc_unpack_frame_sparse.restype = C.c_int64
        L.orc_reduce_frame_l1.argtypes = [u16p, u16p, C.c_uint64, C.c_uint, u8p, u8p, u64p]
        L.orc_reduce_frame_l1.restype = C.c_uint64
        L.orc_lz4f_decode.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
        L.orc_lz4f_decode.restype = C.c_int64
        L.orc_lz4_block_decode.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
        L.orc_lz4_block_decode.restype = C.c_int64
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _c16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


# ---- per-stage restatements -------------------------------------------------------------------
def threshold(dark, eps):
    """A1: recode_writer.py:126-127.  uint8 darks (source_bit_depth <= 8, misc.py:41-49): the sum stays uint8 and wraps mod 2^8 under
    NumPy 2 (numpy restatement; the C loop below is the uint16 case)."""
    if np.asarray(dark).dtype == np.uint8:
        return ((np.asarray(dark).astype(np.uint16) + (int(eps) & 0xFF)) & 0xFF).astype(np.uint8)
    dark = _c16(dark)
    thr = np.empty_like(dark)
    lib().orc_threshold(_p(dark, C.c_uint16), int(eps), dark.size, _p(thr, C.c_uint16))
    return thr


def binarize_l1(frame, thr):
    """A2+A3: recode_writer.py:437,440 -> (bool[ny,nx], uint16[nnz])."""
    frame, thr = _c16(frame), _c16(thr)
    binary = np.empty(frame.shape, np.uint8)
    pix = np.empty(frame.size, np.uint16)
    n = lib().orc_binarize_l1(_p(frame, C.c_uint16), _p(thr, C.c_uint16), frame.size,
                              _p(binary, C.c_uint8), _p(pix, C.c_uint16))
    return binary.astype(bool), pix[:n].copy()


def pack_binary_frame(binary):
    """A4: recode_writer.py:622-634."""
    b = np.ascontiguousarray(binary, dtype=np.uint8).ravel()
    out = np.empty((b.size + 7) // 8, np.uint8)
    lib().orc_pack_binary_frame(_p(b, C.c_uint8), b.size, _p(out, C.c_uint8))
    return out


def bit_pack(vals, d):
    """A5: recode_writer.py:637-652; raw LE bytes when d % 8 == 0 (:463-464)."""
    vals = _c16(vals).ravel()
    if d % 8 == 0 and d == 16:
        return np.frombuffer(vals.tobytes(), np.uint8).copy()
    out = np.empty((vals.size * d + 7) // 8, np.uint8)
    lib().orc_bit_pack(_p(vals, C.c_uint16), vals.size, d, _p(out, C.c_uint8))
    return out


def bit_unpack(packed, n, d):
    packed = np.ascontiguousarray(packed, np.uint8)
    out = np.empty(n, np.uint64)
    lib().orc_bit_unpack(_p(packed, C.c_uint8), n, d, _p(out, C.c_uint64))
    return out


def unpack_frame_sparse(nx, ny, d, bitmap, pix, level=1):
    """A10: reader.h:10-68 -> uint64[nnz,3] (row, col, val)."""
    bitmap = np.ascontiguousarray(bitmap, np.uint8)
    pix = np.ascontiguousarray(pix if pix is not None and len(pix) else np.zeros(1, np.uint8), np.uint8)
    cap = int(np.unpackbits(bitmap, bitorder="little")[: nx * ny].sum())
    out = np.empty((max(cap, 1), 3), np.uint64)
    n = lib().orc_unpack_frame_sparse(nx, ny, d, _p(bitmap, C.c_uint8), _p(pix, C.c_uint8),
                                      _p(out, C.c_uint64), level)
    return out[:n].copy()


def reduce_frame_l1(frame, thr, d):
    """Fused A2..A5 (the timed CPU-baseline form) -> (bitmap u8[ceil(N/8)], packed u8[], nnz)."""
    frame, thr = _c16(frame).ravel(), _c16(thr).ravel()
    bitmap = np.empty((frame.size + 7) // 8, np.uint8)
    packed = np.empty(frame.size * 2 + 8, np.uint8)
    npk = C.c_uint64(0)
    nnz = lib().orc_reduce_frame_l1(_p(frame, C.c_uint16), _p(thr, C.c_uint16), frame.size, d,
                                    _p(bitmap, C.c_uint8), _p(packed, C.c_uint8), C.byref(npk))
    return bitmap, packed[: npk.value].copy(), int(nnz)


def lz4f_decode(data, cap):
    data = np.frombuffer(bytes(data), np.uint8)
    out = np.empty(max(cap, 1), np.uint8)
    n = lib().orc_lz4f_decode(_p(data, C.c_uint8), data.size, _p(out, C.c_uint8), cap)
    if n < 0:
        raise ValueError("orc_lz4f_decode: malformed LZ4 frame (code %d)" % n)
    return out[:n].tobytes()


def lz4_block_decode(data, cap):
    data = np.frombuffer(bytes(data), np.uint8)
    out = np.empty(max(cap, 1), np.uint8)
    n = lib().orc_lz4_block_decode(_p(data, C.c_uint8), data.size, _p(out, C.c_uint8), cap)
    if n < 0:
        raise ValueError("malformed LZ4 block (code %d)" % n)
    return out[:n].tobytes()


def blosc1_decode(chunk):
    """From-spec decoder of a blosc1 chunk (c-blosc 1.x README_CHUNK_FORMAT.rst / blosc.c blosc_d), LZ4 codec only:
    16-byte header (version, versionlz, flags, typesize, nbytes, blocksize, cbytes), `memcpyed` chunks, int32 bstarts,
    per block nsplits x (int32 csize + LZ4 block | stored), then un-shuffle (byte 0x01 / bit 0x04) per block.
    Reference call site whose inverse this is: pyrecode/recode_compressors.py:61-76 (blosc.decompress)."""
    c = bytes(chunk)
    version, versionlz, flags, typesize = c[0], c[1], c[2], c[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<iii", c, 4)
    if version != 2 or cbytes != len(c):
        raise ValueError("bad blosc1 header")
    if flags & 0x02:
        if cbytes != 16 + nbytes:
            raise ValueError("bad memcpyed chunk")
        return c[16:16 + nbytes]
    if (flags >> 5) != 1 or versionlz != 1:
        raise ValueError("not an LZ4 blosc chunk")
    nblocks = -(-nbytes // blocksize)
    bstarts = struct.unpack_from("<%di" % nblocks, c, 16)
    out = bytearray()
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = (not (flags & 0x10)) and typesize <= 16 and blocksize // typesize >= 128 and not leftover
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        pos, tmp = bstarts[b], b""
        for _ in range(nsplits):
            csize, = struct.unpack_from("<i", c, pos)
            pos += 4
            tmp += c[pos:pos + csize] if csize == neblock else lz4_block_decode(c[pos:pos + csize], neblock)
            pos += csize
        if len(tmp) != bsize:
            raise ValueError("block %d decodes to %d bytes, expected %d" % (b, len(tmp), bsize))
        a = np.frombuffer(tmp, np.uint8)
        if flags & 0x04 and bsize >= typesize:       # bit-unshuffle: S elements (multiple of 8), rows of S/8 bytes
            S = (bsize // typesize) & ~7
            if S:
                rows = np.unpackbits(a[:S * typesize].reshape(typesize * 8, S // 8), axis=1, bitorder="little")  # [row r][elem i]
                bits = rows.T.reshape(S, typesize, 8)                                                        # [elem][byte k][bit b]
                a = np.concatenate([np.packbits(bits, axis=2, bitorder="little").reshape(-1), a[S * typesize:]])
        elif flags & 0x01 and typesize > 1:         # byte-unshuffle
            ne = bsize // typesize
            a = np.concatenate([a[:ne * typesize].reshape(typesize, ne).T.reshape(-1), a[ne * typesize:]])
        out += a.tobytes()
    return bytes(out)


# ---- record / file assembly (A7, appendix A of SURVEY.md) ----------------------------------------
# ---- sources beyond 16 bits (uint32 frames: source_bit_depth > 16, pyrecode/misc.py:41-49) - numpy restatements ---------------------------
def threshold32(dark, eps):
    """A1 for uint32 darks: recode_writer.py:126-127; the sum stays uint32 and wraps mod 2^32 under NumPy 2."""
    return ((np.asarray(dark).astype(np.uint64) + (int(eps) & 0xFFFFFFFF)) & 0xFFFFFFFF).astype(np.uint32)


def bit_pack32(vals, d):
    """A5 for uint32 values: recode_writer.py:463-475,637-652 - the low d bits of every value, LSB first, value after value; when d is a
    multiple of 8 the writer takes `.tobytes()` instead: FOUR bytes a value for a uint32 array, whatever d says (24 and 32 alike)."""
    vals = np.ascontiguousarray(vals, dtype=np.uint32)
    if d % 8 == 0:
        return np.frombuffer(vals.astype('<u4').tobytes(), np.uint8)
    bits = np.unpackbits(vals.astype('<u4').view(np.uint8).reshape(-1, 4), axis=1, bitorder='little')[:, :d]
    return np.packbits(bits.reshape(-1), bitorder='little')


def l1_record32(frame, thr, d, frame_id, mode=1, compress=lambda b: zlib.compress(b, 1)):
    """One part-file record for L1 from uint32 frame / threshold (recode_writer.py:437-440,482-525,559-574): as l1_record, in numpy."""
    frame, thr = np.asarray(frame, np.uint32), np.asarray(thr, np.uint32)
    binary = frame > thr
    pix = (frame[binary] - thr[binary]).astype(np.uint32)
    bitmap = np.packbits(binary.reshape(-1), bitorder='little').tobytes()
    packed = bit_pack32(pix, d).tobytes()
    if mode == 0:
        return struct.pack("<II", frame_id, len(packed)) + bitmap + packed, (len(packed),)
    cb, cp = compress(bitmap), compress(packed)
    md = (len(cb), len(cp), len(packed))
    return struct.pack("<IIII", frame_id, *md) + cb + cp, md


def l1_record(frame, thr, d, frame_id, mode=1, compress=lambda b: zlib.compress(b, 1)):
    """One part-file record for L1.  recode_writer.py:482-525, _write_to_frame_buffer :559-574.
    mode 1: u32 frame_id | u32 n_comp_bitmap | u32 n_comp_pix | u32 n_packed_pix | comp_bitmap | comp_pix
    mode 0: u32 frame_id | u32 n_packed_pix | bitmap | packed_pix
    Returns (record bytes, metadata tuple without frame_id)."""
    binary, pix = binarize_l1(frame, thr)
    bitmap = pack_binary_frame(binary).tobytes()
    packed = bit_pack(pix, d).tobytes()
    if mode == 0:
        return struct.pack("<II", frame_id, len(packed)) + bitmap + packed, (len(packed),)
    cb, cp = compress(bitmap), compress(packed)
    md = (len(cb), len(cp), len(packed))
    return struct.pack("<IIII", frame_id, *md) + cb + cp, md


def l3_record(frame, thr, frame_id, mode=1, compress=lambda b: zlib.compress(b, 1)):
    """L3: bitmap only.  recode_writer.py:491-494, 534-550."""
    bitmap = pack_binary_frame(_c16(frame) > _c16(thr)).tobytes()
    if mode == 0:
        return struct.pack("<I", frame_id) + bitmap, ()
    cb = compress(bitmap)
    return struct.pack("<II", frame_id, len(cb)) + cb, (len(cb),)


def node_frames(n_frames, n_nodes, node_id):
    """Contiguous-block ownership rule, recode_writer.py:320-322."""
    per = -(-n_frames // n_nodes)
    lo = node_id * per
    return lo, min(per, max(n_frames - lo, 0))
