"""The oracle (oracle/recode_oracle.c via oracle/oracle.py) against fixtures captured from the reference
itself (tests/golden/make_golden.py).  CPU only.  This is what pins the oracle (task §3)."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, load_npz
from oracle import oracle as orc

G12 = load_npz("g1_g2_reduce.npz")
TAGS = sorted({k.split("_")[1] for k in G12.files if k.startswith("g1_")})


@pytest.mark.parametrize("tag", TAGS)
def test_g1_threshold_binary_residuals(tag):
    dark, eps, frames = G12[f"g1_{tag}_dark"], int(G12[f"g1_{tag}_eps"]), G12[f"g1_{tag}_frames"]
    thr = orc.threshold(dark, eps)
    assert np.array_equal(thr, G12[f"g1_{tag}_thr"])
    for z in range(frames.shape[0]):
        binary, pix = orc.binarize_l1(frames[z], thr)
        assert np.array_equal(binary, G12[f"g1_{tag}_binary{z}"])
        assert np.array_equal(pix, G12[f"g1_{tag}_pix{z}"])


@pytest.mark.parametrize("tag", TAGS)
def test_g2_bitmap_and_packed(tag):
    d = int(G12[f"g1_{tag}_depth"])
    frames, thr = G12[f"g1_{tag}_frames"], G12[f"g1_{tag}_thr"]
    for z in range(frames.shape[0]):
        bm = orc.pack_binary_frame(G12[f"g1_{tag}_binary{z}"])
        assert np.array_equal(bm, G12[f"g2_{tag}_bitmap{z}"])
        pk = orc.bit_pack(G12[f"g1_{tag}_pix{z}"], d)
        assert np.array_equal(pk, G12[f"g2_{tag}_packed{z}"])
        # the fused CPU-baseline form must agree with the staged form
        fbm, fpk, nnz = orc.reduce_frame_l1(frames[z], thr, d)
        assert nnz == len(G12[f"g1_{tag}_pix{z}"])
        assert np.array_equal(fbm, bm) and np.array_equal(fpk, pk)


@pytest.mark.parametrize("d", [1, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16])
def test_g2_bit_pack_all_depths(d):
    vals = G12["g2_vals"]
    out = np.empty((vals.size * d + 7) // 8, np.uint8)
    import ctypes as C
    n = orc.lib().orc_bit_pack(vals.ctypes.data_as(C.POINTER(C.c_uint16)), vals.size, d,
                               out.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert n == out.size
    assert np.array_equal(out, G12[f"g2_bitpack_d{d}"])
    # numpy identity quoted in SURVEY §0.2
    bits = ((vals[:, None] >> np.arange(d)) & 1).astype(np.uint8).ravel()
    assert np.array_equal(out, np.packbits(bits, bitorder="little"))
    back = orc.bit_unpack(out, vals.size, d)
    assert np.array_equal(back, vals.astype(np.uint64) & ((1 << d) - 1))


def test_g5_sparse_expand():
    g5 = load_npz("g5_expand.npz")
    for tag in "abcd":
        ny, nx, d, level = (int(x) for x in g5[f"g5_{tag}_shape"])
        trip = orc.unpack_frame_sparse(nx, ny, d, g5[f"g5_{tag}_bitmap"], g5[f"g5_{tag}_packed"], level)
        assert np.array_equal(trip, g5[f"g5_{tag}_triplets"])


def _parse_part(path, level, mode):
    """SURVEY appendix A part-file layout -> list of raw record bytes."""
    blob = open(path, "rb").read()
    nx, ny = struct.unpack_from("<II", blob, 15)
    nb = (nx * ny + 7) // 8
    pos, recs = 512, []
    while pos < len(blob):
        if level == 1 and mode == 1:
            _, cb, cp, _ = struct.unpack_from("<IIII", blob, pos)
            ln = 16 + cb + cp
        elif level == 1 and mode == 0:
            _, npk = struct.unpack_from("<II", blob, pos)
            ln = 8 + nb + npk
        elif mode == 1:
            _, cb = struct.unpack_from("<II", blob, pos)
            ln = 8 + cb
        else:
            ln = 4 + nb
        recs.append(blob[pos:pos + ln])
        pos += ln
    assert pos == len(blob)
    return recs


@pytest.mark.parametrize("tag,level", [("l1z12", 1), ("l1z16", 1), ("l1ro16", 1), ("l3z", 3)])
def test_g3_records_byte_exact(tag, level):
    g = load_npz(f"g3_{tag}.npz")
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    dark, frames, nodes = g["dark"], g["frames"], int(g["n_nodes"])
    thr = orc.threshold(dark, cfg["calibration_threshold_epsilon"])
    mode, d = cfg["rc_operation_mode"], cfg["source_bit_depth"]
    merged_md, merged_data = [], b""
    for node in range(nodes):
        recs = _parse_part(os.path.join(GOLDEN, "files", "g3_%s.rc%d_part%03d" % (tag, level, node)), level, mode)
        lo, cnt = orc.node_frames(frames.shape[0], nodes, node)
        assert len(recs) == cnt
        for i, ref_rec in enumerate(recs):
            if level == 1:
                rec, md = orc.l1_record(frames[lo + i], thr, d, lo + i, mode)
            else:
                rec, md = orc.l3_record(frames[lo + i], thr, lo + i, mode)
            assert rec == ref_rec
            hdr = 4 + 4 * len(md)
            merged_md.append(struct.pack("<%dI" % len(md), *md))
            merged_data += rec[hdr:]
    # merged-file layout (recode_reader.py:518-592): header | nz x metadata | data blobs
    merged = open(os.path.join(GOLDEN, "files", "g3_%s.rc%d" % (tag, level)), "rb").read()
    assert merged[512:] == b"".join(merged_md) + merged_data
    part0 = open(os.path.join(GOLDEN, "files", "g3_%s.rc%d_part000" % (tag, level)), "rb").read()
    assert merged[:23] == part0[:23] and merged[27:512] == part0[27:512]
    assert struct.unpack_from("<I", merged, 23)[0] == frames.shape[0]
    if level == 1 and g["decoded"].size:
        assert np.array_equal(g["decoded"], np.where(frames > thr, frames - thr, 0).astype(np.uint16))


@pytest.mark.parametrize("tag", ["u8d8", "u8d6", "u8d8v", "u8cast"])
def test_g10_uint8_sources_records_byte_exact(tag):
    """G10: the reference's writer on 8-bit sources (source dtype uint8, misc.py:41-49): uint8 threshold sum, compare and residuals
    (recode_writer.py:126-137,437-440), one byte a value at d = 8 (`.tobytes()`, :463-464), bit-packed at d = 6.  The oracle's uint16
    restatement on the widened frames must give the same records - values and order are the same, only the container type differs -
    and the reference's own reader decodes to where(frame > thr, frame - thr, 0) as uint8."""
    g = load_npz("g10_%s.npz" % tag)
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    dark, frames, nodes = g["dark"], g["frames"], int(g["n_nodes"])
    assert dark.dtype == np.uint8 and frames.dtype == np.uint8 and str(g["decoded_dtype"]) == "uint8"
    thr = orc.threshold(dark, cfg["calibration_threshold_epsilon"])
    assert thr.dtype == np.uint8
    d = cfg["source_bit_depth"]
    merged_md, merged_data = [], b""
    for node in range(nodes):
        recs = _parse_part(os.path.join(GOLDEN, "files", "g10_%s.rc1_part%03d" % (tag, node)), 1, 1)
        lo, cnt = orc.node_frames(frames.shape[0], nodes, node)
        assert len(recs) == cnt
        for i, ref_rec in enumerate(recs):
            rec, md = orc.l1_record(frames[lo + i].astype(np.uint16), thr.astype(np.uint16), d, lo + i, 1)
            assert rec == ref_rec
            assert md[2] == (int((frames[lo + i] > thr).sum()) * d + 7) // 8
            merged_md.append(struct.pack("<3I", *md))
            merged_data += rec[16:]
    merged = open(os.path.join(GOLDEN, "files", "g10_%s.rc1" % tag), "rb").read()
    assert merged[512:] == b"".join(merged_md) + merged_data
    assert np.array_equal(g["decoded"], np.where(frames > thr, frames - thr, 0).astype(np.uint8))


@pytest.mark.parametrize("tag", ["u32d20", "u32d32", "u32d24", "u32d17", "u32d20v"])
def test_g11_uint32_sources_records_byte_exact(tag):
    """G11: the reference's writer on sources beyond 16 bits (uint32 frames, misc.py:41-49): uint32 threshold sum, compare and residuals,
    20- and 17-bit fields through _bit_pack, four raw bytes a value at d = 32 AND d = 24 (`.tobytes()`, recode_writer.py:463-464).  The
    oracle's numpy restatement gives the reference's records; its reader returns where(frame > thr, frame - thr, 0) - except at d = 24,
    where it takes 24-bit fields out of the 32-bit values (the fixture keeps what it returned)."""
    g = load_npz("g11_%s.npz" % tag)
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    dark, frames, nodes = g["dark"], g["frames"], int(g["n_nodes"])
    assert dark.dtype == np.uint32 and frames.dtype == np.uint32 and str(g["decoded_dtype"]) == "uint32"
    thr = orc.threshold32(dark, cfg["calibration_threshold_epsilon"])
    d = cfg["source_bit_depth"]
    merged_md, merged_data = [], b""
    for node in range(nodes):
        recs = _parse_part(os.path.join(GOLDEN, "files", "g11_%s.rc1_part%03d" % (tag, node)), 1, 1)
        lo, cnt = orc.node_frames(frames.shape[0], nodes, node)
        assert len(recs) == cnt
        for i, ref_rec in enumerate(recs):
            rec, md = orc.l1_record32(frames[lo + i], thr, d, lo + i, 1)
            assert rec == ref_rec
            nnz = int((frames[lo + i] > thr).sum())
            assert md[2] == (4 * nnz if d % 8 == 0 else (nnz * d + 7) // 8)
            merged_md.append(struct.pack("<3I", *md))
            merged_data += rec[16:]
    merged = open(os.path.join(GOLDEN, "files", "g11_%s.rc1" % tag), "rb").read()
    assert merged[512:] == b"".join(merged_md) + merged_data
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint64)
    assert np.array_equal(g["decoded"], want) == (d != 24)


def test_g7_stream_mode_records_and_ids():
    """G7: the reference's writer fed chunk after chunk (mode='stream'): every chunk is split by the contiguous-block rule and the ids run
    on from chunk to chunk (recode_writer.py:311-322,383,422) - the oracle's records with those ids are the part files' bytes."""
    g = load_npz("g7_stream.npz")
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    dark, frames, nodes, chunks = g["dark"], g["frames"], int(g["n_nodes"]), g["chunks"].tolist()
    thr = orc.threshold(dark, cfg["calibration_threshold_epsilon"])
    for node in range(nodes):
        ids, at = [], 0
        for c in chunks:
            lo, cnt = orc.node_frames(c, nodes, node)
            ids += [at + lo + i for i in range(cnt)]
            at += c
        assert ids == g["ids_part%d" % node].tolist()
        recs = _parse_part(os.path.join(GOLDEN, "files", "g7_stream.rc1_part%03d" % node), 1, 1)
        assert len(recs) == len(ids)
        for fid, ref_rec in zip(ids, recs):
            assert orc.l1_record(frames[fid], thr, cfg["source_bit_depth"], fid, 1)[0] == ref_rec


@pytest.mark.parametrize("tag,level", [("l1bz2", 1), ("l1lzma", 1), ("l1z12_lvl9", 1), ("l3ro", 3)])
def test_g9_other_host_schemes_records_byte_exact(tag, level):
    """G9: the reference's part files for bz2, lzma, zlib level 9 and level 3 without compression = the oracle's pieces through the same
    standard-library call (recode_compressors.py:82-101)."""
    import bz2
    import lzma
    import zlib
    g = load_npz("g9_%s.npz" % tag)
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    dark, frames, nodes = g["dark"], g["frames"], int(g["n_nodes"])
    thr = orc.threshold(dark, cfg["calibration_threshold_epsilon"])
    mode, d, lvl = cfg["rc_operation_mode"], cfg["source_bit_depth"], cfg["compression_level"]
    comp = {0: lambda b: zlib.compress(b, lvl), 4: lambda b: bz2.compress(b, compresslevel=lvl), 5: lambda b: lzma.compress(b, preset=lvl)}[cfg["compression_scheme"]]
    for node in range(nodes):
        recs = _parse_part(os.path.join(GOLDEN, "files", "g9_%s.rc%d_part%03d" % (tag, level, node)), level, mode)
        lo, cnt = orc.node_frames(frames.shape[0], nodes, node)
        assert len(recs) == cnt
        for i, ref_rec in enumerate(recs):
            rec = orc.l1_record(frames[lo + i], thr, d, lo + i, mode, comp)[0] if level == 1 else orc.l3_record(frames[lo + i], thr, lo + i, mode, comp)[0]
            assert rec == ref_rec


def test_against_reference_c_loops_when_built():
    """oracle/_ref/libreader_ref.so = the reference's own reader.h compiled in place (build_ref.sh)."""
    import ctypes as C
    path = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libreader_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    ref = C.CDLL(path)
    ref._unpack_frame_sparse.restype = C.c_int64
    ref._unpack_frame_sparse.argtypes = [C.c_uint16, C.c_uint16, C.c_uint8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint8]
    ref._bit_pack_pixel_intensities.restype = C.c_float
    ref._bit_pack_pixel_intensities.argtypes = [C.c_uint64, C.c_uint32, C.c_uint8, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9)
    for ny, nx, d, s in [(33, 47, 12, 0.2), (64, 64, 16, 0.01), (10, 130, 9, 0.5), (8, 8, 13, 1.0), (8, 8, 12, 0.0)]:
        binary = rng.random((ny, nx)) < s
        vals = rng.integers(0, 65536, int(binary.sum())).astype(np.uint16)
        mine = orc.bit_pack(vals, d) if d != 16 else orc.bit_pack(vals, 16)
        theirs = np.full(max((vals.size * d + 7) // 8, 1), 0xAA, np.uint8)  # dirty buffer: zeroing is part of the spec
        ref._bit_pack_pixel_intensities((vals.size * d + 7) // 8, vals.size, d, vals.ctypes.data, theirs.ctypes.data)
        assert np.array_equal(mine, theirs[: mine.size])
        bitmap = np.packbits(binary.ravel(), bitorder="little")
        buf = np.zeros((max(vals.size, 1), 3), np.uint64)
        pk = np.concatenate([mine, np.zeros(8, np.uint8)])
        n = ref._unpack_frame_sparse(nx, ny, d, bitmap.ctypes.data, pk.ctypes.data, buf.ctypes.data, 1)
        got = orc.unpack_frame_sparse(nx, ny, d, bitmap, pk, 1)
        assert n == vals.size == got.shape[0]
        assert np.array_equal(got, buf[:n])


def test_blosc1_from_spec_decoder_on_a_hand_built_chunk():
    """A chunk assembled by hand from the format description: bit-shuffled stored blocks, typesize 8, plus a memcpyed one."""
    import struct
    rng = np.random.default_rng(1)
    data = rng.integers(0, 256, 512 + 200, dtype=np.uint8)
    blocks = []
    for lo in (0, 512):
        blk = data[lo:lo + 512]
        S = (blk.size // 8) & ~7
        bits = np.unpackbits(blk[:S * 8].reshape(S, 8), axis=1, bitorder="little")          # [elem][bit r]
        rows = np.packbits(bits.T, axis=1, bitorder="little").reshape(-1)                   # [row r][S/8 bytes]
        sh = np.concatenate([rows, blk[S * 8:]]).tobytes()
        blocks.append(struct.pack("<i", len(sh)) + sh)                                      # csize == size -> stored
    tab = 16 + 8
    bstarts = [tab, tab + len(blocks[0])]
    body = struct.pack("<2i", *bstarts) + b"".join(blocks)
    chunk = bytes([2, 1, 0x34, 8]) + struct.pack("<iii", data.size, 512, 16 + len(body)) + body
    assert orc.blosc1_decode(chunk) == data.tobytes()
    mem = bytes([2, 1, 0x36, 8]) + struct.pack("<iii", 5, 5, 21) + b"hello"
    assert orc.blosc1_decode(mem) == b"hello"


def test_blosc1_from_spec_decoder_on_compressed_bitshuffled_blocks():
    """The path the device encoder actually emits for scheme 8: bit-shuffled blocks that ARE LZ4-compressed.  Block 0's LZ4
    bytes are derived by hand from lz4_Block_format.md (an all-zero shuffled block: one literal, one 506-byte overlapping
    match, the five mandatory trailing literals); block 1 is compressed by the stock liblz4 block API when the library is
    installed (an encoder that shares nothing with this repository)."""
    import ctypes as C
    import ctypes.util
    import struct
    rng = np.random.default_rng(4)

    def shuffle(blk):
        S = (blk.size // 8) & ~7
        bits = np.unpackbits(blk[:S * 8].reshape(S, 8), axis=1, bitorder="little")
        return np.concatenate([np.packbits(bits.T, axis=1, bitorder="little").reshape(-1), blk[S * 8:]])

    blk0 = np.zeros(512, np.uint8)
    lz0 = bytes([0x1F, 0x00, 0x01, 0x00, 0xFF, 0xE8, 0x50, 0, 0, 0, 0, 0])   # token(1 lit, ml 15+) 00 | off 1 | +255 +232 | token(5 lit) 00*5
    assert orc.lz4_block_decode(lz0, 512) == bytes(512)
    blk1 = np.zeros(512, np.uint8)                      # sparse, like a packed binary map: a few single-bit bytes
    blk1[rng.choice(512, 20, replace=False)] = 1 << rng.integers(0, 8, 20).astype(np.uint8)
    sh1 = shuffle(blk1).tobytes()
    name = ctypes.util.find_library("lz4")
    if name:
        L = C.CDLL(name)
        L.LZ4_compress_default.restype = C.c_int
        L.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        dst = C.create_string_buffer(1024)
        n = L.LZ4_compress_default(sh1, dst, len(sh1), 1024)
        assert 0 < n < 512
        lz1 = dst.raw[:n]
    else:
        lz1 = sh1                                        # stored (csize == size)
    blocks = [struct.pack("<i", len(lz0)) + lz0, struct.pack("<i", len(lz1)) + lz1]
    tab = 16 + 8
    body = struct.pack("<2i", tab, tab + len(blocks[0])) + b"".join(blocks)
    chunk = bytes([2, 1, 0x34, 8]) + struct.pack("<iii", 1024, 512, 16 + len(body)) + body
    assert orc.blosc1_decode(chunk) == blk0.tobytes() + blk1.tobytes()


def test_v01_header_written_by_the_reference_loads():
    """G6: a 321-byte version-0.1 header serialised by the reference's ReCoDeHeader(version=0.1) (tests/golden/make_golden.py
    g6) must load with every field the reference stored (reference recode_header.py:27-56, 188-249)."""
    from pyrecode_amd.recode_header import ReCoDeHeader
    g = load_npz("g6_header_v01.npz")
    path = os.path.join(GOLDEN, "files", "g6_header_v01.bin")
    assert os.path.getsize(path) == 321
    h = ReCoDeHeader()
    h.load(path)
    d = h.as_dict()
    assert h.recode_header_length == 321
    for k, v in zip(g["keys"].tolist(), g["vals"].tolist()):
        assert int(d[k]) == v, k
    assert d["version_major"] == 0 and d["version_minor"] == 1 and d["nx"] == 56 and d["ny"] == 40 and d["nz"] == 7
    assert str(d["source_file_name"]).strip() == str(g["source_file_name"]).strip()
    assert str(d["calibration_file_name"]).strip() == str(g["calibration_file_name"]).strip()
    # what a v0.1 file implies for the fields it does not carry (reference :236-243)
    assert d["is_bit_packed"] == 1 and d["source_header_length"] == 0 and d["source_dtype"] == 0 and d["target_dtype"] == 0
    # and the table round-trips byte for byte
    assert h.to_bytes() == open(path, "rb").read()


def test_committed_fixtures_are_what_the_reference_writes_today(tmp_path):
    """The pin itself: where the reference is present (this container, never the GPU box), tests/golden/make_golden.py is run
    again into a scratch directory and everything it writes is compared with the committed set - files byte for byte, the arrays of
    every .npz one by one.  A fixture edited by hand, a generator that drifted from its fixtures, or a reference that changed under the
    generator would show here."""
    import subprocess
    import sys
    ref = os.environ.get("RECODE_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "pyrecode")):
        pytest.skip("the reference is not present here (the GPU box): the fixtures cannot be regenerated")
    out = str(tmp_path / "golden")
    os.makedirs(out)
    env = dict(os.environ, RC_GOLDEN_OUT=out)
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    made_files = sorted(os.listdir(os.path.join(out, "files")))
    have_files = sorted(f for f in os.listdir(os.path.join(GOLDEN, "files")) if not f.startswith("recode_params_"))   # (the reference's own parameter file: data it ships, not an output)
    assert made_files == have_files
    for fn in made_files:
        a = open(os.path.join(out, "files", fn), "rb").read()
        b = open(os.path.join(GOLDEN, "files", fn), "rb").read()
        assert a == b, "files/%s differs from what the reference writes now" % fn
    made_npz = sorted(f for f in os.listdir(out) if f.endswith(".npz"))
    have_npz = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert made_npz == have_npz
    for fn in made_npz:
        with np.load(os.path.join(out, fn), allow_pickle=False) as a, np.load(os.path.join(GOLDEN, fn), allow_pickle=False) as b:
            assert sorted(a.files) == sorted(b.files), fn
            for k in a.files:
                assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), "%s[%s] differs" % (fn, k)
