"""-m gpu: compression_scheme 0 on the device (RC_SCHEME_ZLIB_DEVICE, rc_deflate_block.h / rc_deflate.hip) through the C ABI.  Every stream
must be a zlib stream that STDLIB zlib - the decoder the reference's reader calls (pyrecode/recode_compressors.py:43) - expands to the
oracle's bytes, its length must equal the record's metadata, and the device's bytes must equal the serial model's
(tests/deflate_block_model.py, judged by zlib on the CPU) tile for tile."""
import struct
import zlib

import numpy as np
import pytest

import deflate_block_model as model
from conftest import synth_frames
from test_gpu_parity import SHAPES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pyrecode_amd import _lib
    if _lib.device_count() == 0:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return _lib


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def _check_record(orc, r, frame, thr, d, fid, md_row=None, level=1):
    binary = frame > thr
    bitmap = orc.pack_binary_frame(binary).tobytes()
    if level == 3:
        got_fid, cb = struct.unpack_from("<II", r, 0)
        assert got_fid == fid and len(r) == 8 + cb
        assert zlib.decompress(r[8:]) == bitmap
        assert r[8:] == model.bitmap_stream(bitmap)
        return
    got_fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
    assert got_fid == fid and len(r) == 16 + cb + cp
    if md_row is not None:
        assert (cb, cp, npk) == tuple(int(v) for v in md_row)
    _, pix = orc.binarize_l1(frame, thr)
    packed = orc.bit_pack(pix, d).tobytes()
    assert npk == len(packed)
    assert zlib.decompress(r[16:16 + cb]) == bitmap           # (zlib.decompress also checks the Adler-32 and that nothing trails the stream)
    assert zlib.decompress(r[16 + cb:]) == packed
    assert r[16:16 + cb] == model.bitmap_stream(bitmap), "binary-map stream differs from the serial model"
    assert r[16 + cb:] == model.stored_stream(packed), "residual stream differs from the serial model"


@pytest.mark.parametrize("ny,nx,s,d,eps", SHAPES)
def test_device_zlib_records_inflate_bit_exact(hip, orc, ny, nx, s, d, eps):
    dark, frames = synth_frames(41 + nx, 4, ny, nx, s, d)
    thr = orc.threshold(dark, eps)
    ctx = hip.ReduceContext(nx, ny, d, 1, 1, 0, 1, 0, max_batch=4, device_zlib=True)
    assert ctx.on_device_codec
    ctx.set_dark(dark, eps)
    out, rec, md = ctx.reduce_compress_batch(frames, first_frame_id=7)
    for z in range(frames.shape[0]):
        _check_record(orc, out[int(rec[z]):int(rec[z + 1])].tobytes(), frames[z], thr, d, 7 + z, md[z])
    ctx.close()


def test_device_zlib_special_tiles_equal_the_model(hip, orc):
    """Frames whose maps ARE the format tests' special blocks (events at the block's ends, periodic units, growing gaps, two-bit bytes, the
    stored fallback, matches of more than 258 bytes): one 512-byte block per tile, through the fused kernel."""
    from test_lz4_format_cpu import _blocks
    blocks = [b for b in _blocks() if len(b) == 512]
    nt = len(blocks)
    bits = np.unpackbits(np.frombuffer(b"".join(blocks), np.uint8), bitorder="little").astype(bool)
    ny, nx = nt * 8, 512                      # a tile = 8 rows of 512 pixels
    frame = np.where(bits.reshape(ny, nx), 1000, 0).astype(np.uint16)
    frames = np.stack([frame, np.zeros_like(frame), np.roll(frame, 4096 * 3 + 17)])
    dark = np.full((ny, nx), 100, np.uint16)
    thr = orc.threshold(dark, 0)
    ctx = hip.ReduceContext(nx, ny, 12, 1, 1, 0, 1, 0, max_batch=3, device_zlib=True)
    ctx.set_dark(dark, 0)
    out, rec, md = ctx.reduce_compress_batch(frames, first_frame_id=0)
    for z in range(3):
        _check_record(orc, out[int(rec[z]):int(rec[z + 1])].tobytes(), frames[z], thr, 12, z, md[z])
    ctx.close()


@pytest.mark.parametrize("d", [1, 3, 8, 11, 16])
def test_device_zlib_depths_and_stored_block_borders(hip, orc, d):
    """The residual stream crosses several 32 KiB stored-block borders at bit phases of every kind."""
    ny, nx = 512, 700
    dark, frames = synth_frames(900 + d, 2, ny, nx, 0.22, 12)
    thr = orc.threshold(dark, 0)
    ctx = hip.ReduceContext(nx, ny, d, 1, 1, 0, 1, 0, max_batch=2, device_zlib=True)
    ctx.set_dark(dark, 0)
    out, rec, md = ctx.reduce_compress_batch(frames, first_frame_id=0)
    for z in range(2):
        _check_record(orc, out[int(rec[z]):int(rec[z + 1])].tobytes(), frames[z], thr, d, z, md[z])
    ctx.close()


def test_device_zlib_level3_and_edge_frames(hip, orc):
    ny, nx = 200, 333
    dark, frames = synth_frames(77, 4, ny, nx, 0.02, 12)
    frames[1] = 0                        # nothing set: the residual stream is one empty stored block
    frames[2] = 4000                     # everything set: stored tiles (the record still fits: d = 12)
    thr = orc.threshold(dark, 0)
    for level in (1, 3):
        ctx = hip.ReduceContext(nx, ny, 12, level, 1, 0, 1, 0, max_batch=4, device_zlib=True)
        ctx.set_dark(dark, 0)
        out, rec, md = ctx.reduce_compress_batch(frames, first_frame_id=50)
        for z in range(4):
            _check_record(orc, out[int(rec[z]):int(rec[z + 1])].tobytes(), frames[z], thr, 12, 50 + z, md[z] if level == 1 else None, level)
        ctx.close()


def test_device_zlib_uint8_sources_and_level2(hip, orc):
    ny, nx = 130, 260
    rng = np.random.default_rng(3)
    dark = rng.integers(8, 12, (ny, nx)).astype(np.uint8)
    frames = np.where(rng.random((3, ny, nx)) < 0.03, rng.integers(20, 250, (3, ny, nx)), rng.integers(0, 8, (3, ny, nx))).astype(np.uint8)
    thr = dark.astype(np.uint16)
    ctx = hip.ReduceContext(nx, ny, 8, 1, 1, 0, 1, 0, max_batch=3, src_dtype=np.uint8, device_zlib=True)
    ctx.set_dark(dark, 0)
    out, rec, md = ctx.reduce_compress_batch(frames, first_frame_id=0)
    for z in range(3):
        _check_record(orc, out[int(rec[z]):int(rec[z + 1])].tobytes(), frames[z].astype(np.uint16), thr, 8, z, md[z])
    ctx.close()
    # level 2: the statistics take the residuals' place, framed exactly like level 1 - compare with the LZ4 ctx's streams after decoding
    dark16, frames16 = synth_frames(5, 3, ny, nx, 0.02, 12)
    ref = hip.ReduceContext(nx, ny, 12, 2, 0, 0, 1, 0, max_batch=3)
    ref.set_dark(dark16, 0)
    o0, r0, _ = ref.reduce_compress_batch(frames16, first_frame_id=0)
    ctx = hip.ReduceContext(nx, ny, 12, 2, 1, 0, 1, 0, max_batch=3, device_zlib=True)
    ctx.set_dark(dark16, 0)
    out, rec, md = ctx.reduce_compress_batch(frames16, first_frame_id=0)
    nb = (ny * nx + 7) // 8
    for z in range(3):
        plain = o0[int(r0[z]):int(r0[z + 1])].tobytes()          # mode-0 record: id | n_packed | map | statistics
        r = out[int(rec[z]):int(rec[z + 1])].tobytes()
        fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        assert npk == struct.unpack_from("<I", plain, 4)[0]
        assert zlib.decompress(r[16:16 + cb]) == plain[8:8 + nb]
        assert zlib.decompress(r[16 + cb:]) == plain[8 + nb:]
    ref.close()
    ctx.close()


def test_device_zlib_async_pipelined_batches(hip, orc):
    import torch
    ny, nx, B = 256, 1024, 6
    dark, frames = synth_frames(12, 3 * B, ny, nx, 0.015, 14)
    thr = orc.threshold(dark, 0)
    ctx = hip.ReduceContext(nx, ny, 14, 1, 1, 0, 1, 0, max_batch=B, device_zlib=True)
    ctx.set_dark(dark, 0)
    ctx.keep_binary_maps(False)
    ctx.set_pipelined(True)
    fd = torch.from_numpy(frames.view(np.int16)).cuda()
    cap = int(ctx.out_capacity(B))
    outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(3)]
    recs = [torch.empty(B + 1, dtype=torch.int64, device="cuda") for _ in range(3)]
    mds = [torch.empty((B, 3), dtype=torch.int32, device="cuda") for _ in range(3)]
    for i in range(3):
        ctx.enqueue(fd[i * B:(i + 1) * B].data_ptr(), B, i * B, outs[i].data_ptr(), cap, recs[i].data_ptr(), mds[i].data_ptr())
    ctx.sync()
    for i in range(3):
        rec, out, md = recs[i].cpu().numpy(), outs[i].cpu().numpy(), mds[i].cpu().numpy()
        for z in range(B):
            _check_record(orc, out[int(rec[z]):int(rec[z + 1])].tobytes(), frames[i * B + z], thr, 14, i * B + z, md[z])
    ctx.close()


def test_device_zlib_refuses_uint32_sources(hip):
    with pytest.raises(NotImplementedError):
        hip.ReduceContext(64, 64, 24, 1, 1, 0, 1, 0, max_batch=2, src_dtype=np.uint32, device_zlib=True)
    # ... while the host-zlib form of the same ctx exists
    hip.ReduceContext(64, 64, 24, 1, 1, 0, 1, 0, max_batch=2, src_dtype=np.uint32).close()


def test_writer_with_device_zlib_writes_files_the_reference_reader_reads(hip, orc, tmp_path):
    """ReCoDeWriter(device_zlib=True) on the reference's own test configuration (config/recode_params_minimal_read_write_test.txt: L1, zlib,
    d = 12; tests/minimal_read_write_test.py's data): scheme 0 in the header, part files merge, the reader's zlib.decompress path returns the
    frames; the DEFAULT writer still makes the host call (byte-identical files are pinned by the golden tests)."""
    import os
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_writer import ReCoDeWriter
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    here = os.path.dirname(os.path.abspath(__file__))
    rng = np.random.default_rng(0)
    data = np.clip(rng.integers(0, 4096, (9, 512, 512)).astype(np.int32) - 3500, 0, None).astype(np.uint16)
    dark = np.zeros((512, 512), np.uint16)
    sizes = {}
    for dz in (True, False):
        d = tmp_path / ("dz%d" % dz)
        d.mkdir()
        for node in range(3):
            ip = InputParams()
            ip.load(os.path.join(here, "golden", "files", "recode_params_minimal_read_write_test.txt"))
            ip.nx, ip.ny, ip.nz = 512, 512, 9             # as the reference's test does (tests/minimal_read_write_test.py:36-40)
            ip.source_data_type = ip.target_data_type = 0
            w = ReCoDeWriter("t", dark_data=dark, output_directory=str(d), input_params=ip, node_id=node, device_zlib=dz)
            w.start()
            assert w._host_compress == (not dz)
            w.run(data)
            w.close()
        merge_parts(str(d), "t.rc1", 3)
        rd = ReCoDeReader(str(d / "t.rc1"))
        rd.open(print_header=False)
        assert rd.get_header().as_dict()["compression_scheme"] == 0
        for z in range(9):
            assert np.array_equal(np.asarray(rd.get_frame(z)[z]["data"].todense()), data[z])
        rd.close()
        sizes[dz] = os.path.getsize(str(d / "t.rc1"))
    assert sizes[True] < 1.5 * sizes[False], sizes     # (14.5 % density: stored tiles and stored residuals against stock zlib's dynamic blocks)
