"""CPU: the host-side mirror of the reference interface (header, structures, params, reader parsing, merge_parts) against
files written by the reference itself, and the C-ABI library's symbol table.  No compute entry point is called."""
import os
import re
import shutil
import struct

import numpy as np
import pytest

from conftest import GOLDEN, REPO, load_npz

FILES = os.path.join(GOLDEN, "files")
CASES = [("l1z12", 1, 3), ("l1z16", 1, 2), ("l1ro16", 1, 2), ("l3z", 3, 2)]


def test_library_exports_every_declared_symbol():
    from pyrecode_amd import _lib
    hdr = open(os.path.join(REPO, "include", "recode_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", hdr))
    assert declared and declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()  # resolves every symbol; raises AttributeError on a mismatch
    assert L.rc_abi_version() == 3
    assert L.rc_scheme_on_device(2) == 1 and L.rc_scheme_on_device(0) == 0
    assert L.rc_strerror(-5).decode() == "Buffer size smaller than compressed data size"


def test_ctx_arguments_are_checked_before_any_device_is_touched():
    """rc_ctx_create refuses what no launch could take - with or without a GPU: zero sizes, a batch beyond a launch's grid.y, depths beyond 32."""
    from pyrecode_amd import _lib
    for kw, pat in ((dict(max_batch=0), "max_batch"), (dict(max_batch=65536), "65535"), (dict(nx=0), "nx"), (dict(depth=33), "source_bit_depth")):
        a = dict(nx=64, ny=64, depth=12, max_batch=4)
        a.update(kw)
        with pytest.raises((_lib.RecodeHipError, ValueError, NotImplementedError), match=pat):
            _lib.ReduceContext(a["nx"], a["ny"], a["depth"], max_batch=a["max_batch"])


def test_no_gpu_means_loud_failure_not_fallback():
    from pyrecode_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.RecodeHipError, match="no CPU path"):
        _lib.ReduceContext(64, 64, 12)
    out = np.zeros(8, np.uint8)
    vals = np.arange(4, dtype=np.uint16)
    with pytest.raises(_lib.RecodeHipError):
        _lib.check(_lib.lib().rc_bit_pack(vals.ctypes.data, 4, 12, out.ctypes.data, 6))
    # every other stateless compute entry point: RC_ERR_DEVICE, nothing computed on the host
    import ctypes as C
    L = _lib.lib()
    n = C.c_uint64(0)
    src = np.zeros(64, np.uint8)
    big = np.zeros(4096, np.uint8)
    u64 = np.zeros(64, np.uint64)
    sizes = np.array([[8, 8, 8]], np.uint32)
    for scheme in (1, 2, 8):
        assert L.rc_compress(scheme, 1, src.ctypes.data, src.size, big.ctypes.data, big.size, C.byref(n)) == _lib.RC_ERR_DEVICE
        assert L.rc_decompress(scheme, src.ctypes.data, src.size, big.ctypes.data, big.size, C.byref(n)) == _lib.RC_ERR_DEVICE
    assert L.rc_bit_unpack(src.ctypes.data, 6, 4, 12, u64.ctypes.data) == _lib.RC_ERR_DEVICE
    assert L.rc_unpack_frame_sparse(8, 8, 12, src.ctypes.data, src.ctypes.data, 6, u64.ctypes.data, 16, 1) == _lib.RC_ERR_DEVICE
    assert L.rc_expand_frames(8, 8, 12, 1, 0, 0, src.ctypes.data, sizes.ctypes.data, 1, u64.ctypes.data, u64.ctypes.data + 64, 4) == _lib.RC_ERR_DEVICE
    assert L.rc_expand_frames_submit(0, 8, 8, 12, 1, 0, 0, src.ctypes.data, sizes.ctypes.data, 1, u64.ctypes.data, 4) == _lib.RC_ERR_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "pyrecode_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "librecode_oracle" not in text, f


@pytest.mark.parametrize("tag,level,nodes", CASES)
def test_header_roundtrip_and_offsets(tag, level, nodes):
    from pyrecode_amd.recode_header import ReCoDeHeader
    path = os.path.join(FILES, "g3_%s.rc%d_part000" % (tag, level))
    raw = open(path, "rb").read(512)
    h = ReCoDeHeader()
    h.load(path)
    assert h.recode_header_length == 512 and h.to_bytes() == raw
    d = h.as_dict()
    # SURVEY §8f-N3 offsets
    assert struct.unpack_from("<Q", raw, 0)[0] == 158966344846346 == d["uid"]
    assert (raw[8], raw[9], raw[10], raw[11], raw[12], raw[13]) == (0, 2, 1, level, d["rc_operation_mode"], 1)
    assert struct.unpack_from("<III", raw, 15) == (d["nx"], d["ny"], d["nz"])
    assert h.get_field_position_in_bytes("nz") == 23 and h.get_field_position_in_bytes("source_file_name") == 37
    assert h.get_field_position_in_bytes("source_bit_depth") == 258 and raw[258] == d["source_bit_depth"]
    assert d["source_file_name"] == ("g3_" + tag).ljust(100)
    assert h.get_frame_data_offset(True, 12) == 512 and h.get_frame_data_offset(False, 12) == 512 + 12 * d["nz"]


@pytest.mark.parametrize("tag,level,nodes", CASES)
def test_header_create_matches_reference_bytes(tag, level, nodes, tmp_path):
    from pyrecode_amd.params import InitParams, InputParams
    from pyrecode_amd.recode_header import ReCoDeHeader
    g = load_npz("g3_%s.npz" % tag)
    cfg = tmp_path / "p.txt"
    cfg.write_text("".join("%s = %d\n" % (k, int(v)) for k, v in zip(g["cfg_keys"].tolist(), g["cfg_vals"])))
    ip = InputParams()
    ip.load(str(cfg))
    assert ip.validate()
    init = InitParams("batch", str(tmp_path), image_filename="g3_" + tag)
    h = ReCoDeHeader()
    h.create(init, ip, True)
    h.set("source_header_length", 0)
    assert h.validate()
    # the part header's nz is rewritten at close() with the frames that node wrote
    ref = open(os.path.join(FILES, "g3_%s.rc%d_part000" % (tag, level)), "rb").read(512)
    h.update("nz", struct.unpack_from("<I", ref, 23)[0])
    assert h.to_bytes() == ref


def test_params_file_of_the_reference_test_loads():
    from pyrecode_amd.params import InputParams
    ip = InputParams()
    ip.load(os.path.join(FILES, "recode_params_minimal_read_write_test.txt"))
    ip.source_data_type = 0
    ip.target_data_type = 0
    assert ip.validate()
    assert (ip.nx, ip.ny, ip.nz, ip.num_threads) == (512, 512, 9, 3)
    assert (ip.reduction_level, ip.rc_operation_mode, ip.compression_scheme, ip.compression_level) == (1, 1, 0, 1)
    assert ip.source_bit_depth == 12 and ip.source_numpy_dtype is np.uint16
    bad = InputParams()
    with pytest.raises(AssertionError, match="Unknown parameter"):
        bad._param_map.pop("num_rows")
        bad.load(os.path.join(FILES, "recode_params_minimal_read_write_test.txt"))


def test_structures_match_reference_layout():
    from pyrecode_amd.structures import ReCoDeStructures
    s = ReCoDeStructures({"nx": 53, "ny": 37})
    assert s.binary_image_sz_bytes == 246
    assert [f["name"] for f in s.standard_frame_metadata_structure_for(1, 1)] == [
        "bytes_in_compressed_binary_map", "bytes_in_compressed_pixvals", "bytes_in_packed_pixvals"]
    assert s.get_standard_frame_metadata_size(1, 1) == 12 and s.get_standard_frame_metadata_size(1, 0) == 4
    assert s.get_standard_frame_metadata_size(3, 1) == 4 and s.get_standard_frame_metadata_size(3, 0) == 0
    md = {"bytes_in_compressed_binary_map": 10, "bytes_in_compressed_pixvals": 20, "bytes_in_packed_pixvals": 99}
    assert s.get_frame_data_size(1, 1, md) == 30
    assert s.get_frame_data_size(1, 0, {"bytes_in_packed_pixvals": 7}) == 246 + 7
    assert s.get_frame_data_size(3, 0, {}) == 246 and s.get_frame_data_size(4, 1, md) == 10


@pytest.mark.parametrize("tag,level,nodes", CASES)
def test_merge_parts_reproduces_reference_merged_file(tag, level, nodes, tmp_path):
    from pyrecode_amd.recode_reader import merge_parts
    base = "g3_%s.rc%d" % (tag, level)
    for i in range(nodes):
        shutil.copy(os.path.join(FILES, "%s_part%03d" % (base, i)), tmp_path)
    merge_parts(str(tmp_path), base, nodes)
    assert (tmp_path / base).read_bytes() == open(os.path.join(FILES, base), "rb").read()


@pytest.mark.parametrize("tag,level,nodes", CASES)
def test_reader_raw_access_and_seek_table(tag, level, nodes):
    from pyrecode_amd.recode_reader import ReCoDeReader
    g = load_npz("g3_%s.npz" % tag)
    nz = g["frames"].shape[0]
    base = os.path.join(FILES, "g3_%s.rc%d" % (tag, level))
    rd = ReCoDeReader(base, is_intermediate=False)
    rd.open(print_header=False)
    assert rd.get_shape() == (nz, g["frames"].shape[1], g["frames"].shape[2])
    merged = open(base, "rb").read()
    start = 512 + nz * rd.sz_frame_metadata
    pieces = []
    for z in range(nz):
        f = rd.get_next_frame_raw()
        (fid, body), = f.items()
        assert fid == z
        blob = b"".join(body["data"].values())
        assert merged[start + int(rd._seek_table[z, 1]):start + int(rd._seek_table[z, 1]) + len(blob)] == blob
        pieces.append(blob)
    assert b"".join(pieces) == merged[start:]
    rd.close()
    # intermediate file: frame ids follow the contiguous-block rule; EOF -> None
    part = ReCoDeReader(base + "_part001", is_intermediate=True)
    part.open(print_header=False)
    ids = []
    while True:
        f = part.get_next_frame_raw(read_data=False)
        if f is None:
            break
        ids.append(int(list(f.keys())[0]))
    per = -(-nz // nodes)
    assert ids == list(range(per, min(2 * per, nz)))
    with pytest.raises(ValueError):
        part.get_frame(0)
    part.close()
    # the batched readers' index of a part file (one walk over the record headers): same ids, metadata rows and data positions as the
    # sequential walk; a file cut in the middle of its last record indexes the whole records only; the sequential cursor is left alone
    for node in range(nodes):
        path = base + "_part%03d" % node
        part = ReCoDeReader(path, is_intermediate=True)
        part.open(print_header=False)
        seq = []
        while True:
            f = part.get_next_frame_raw(read_data=False)
            if f is None:
                break
            (fid, body), = f.items()
            seq.append((int(fid), {k: int(v) for k, v in body["metadata"].items()}, part.get_file_position()))
        part.close()
        part = ReCoDeReader(path, is_intermediate=True)
        part.open(print_header=False)
        first = part.get_next_frame_raw(read_data=False)
        here = part.get_file_position()
        assert part._batch_frames() == len(seq) and part.get_file_position() == here
        assert part.part_frame_ids.tolist() == [q[0] for q in seq]
        for z, (fid, md, end) in enumerate(seq):
            assert {k: int(v) for k, v in part._frame_metadata[z].items()} == md
            assert part._frame_data_start_position + int(part._seek_table[z, 1]) + int(part._seek_table[z, 0]) == end
        part.close()
    whole = open(base + "_part000", "rb").read()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cut = os.path.join(tmp, "cut.rc%d_part000" % level)
        with open(cut, "wb") as f:
            f.write(whole[:-3])
        part = ReCoDeReader(cut, is_intermediate=True)
        part.open(print_header=False)
        full = ReCoDeReader(base + "_part000", is_intermediate=True)
        full.open(print_header=False)
        assert part._batch_frames() == full._batch_frames() - 1
        part.close()
        full.close()


def test_writer_constructor_validation(tmp_path):
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g3_l1z12.npz")
    cfg = tmp_path / "p.txt"
    cfg.write_text("".join("%s = %d\n" % (k, int(v)) for k, v in zip(g["cfg_keys"].tolist(), g["cfg_vals"])))

    def params(**over):
        ip = InputParams()
        ip.load(str(cfg))
        for k, v in over.items():
            ip._param_map[k] = v
        return ip
    with pytest.raises(RuntimeError, match="different shapes"):
        ReCoDeWriter("x", dark_data=np.zeros((3, 3), np.uint16), output_directory=str(tmp_path), input_params=params())
    with pytest.raises(ValueError, match="Invalid input params"):
        ReCoDeWriter("x", dark_data=g["dark"], output_directory=str(tmp_path), input_params=params(reduction_level=7))
    with pytest.raises(ValueError, match="Invalid initialization parameters"):
        ReCoDeWriter("x", dark_data=g["dark"], output_directory="", input_params=params())
    with pytest.raises(NotImplementedError):
        ReCoDeWriter("x", dark_data=g["dark"], output_directory=str(tmp_path), input_params=params(reduction_level=4))
    w = ReCoDeWriter("x", dark_data=g["dark"], output_directory=str(tmp_path), input_params=params())
    assert w._header["nx"] == 56 and w._header["is_intermediate"] is True


def test_compressor_seam_host_schemes_and_errors():
    from pyrecode_amd import recode_compressors as rcmp
    data = bytes(range(256)) * 8
    for scheme in (0, 4, 5):
        c = rcmp.compress(scheme, 1, data, None)
        assert rcmp.de_compress(scheme, c, None) == data
    import zlib
    assert rcmp.compress(0, 1, data, None) == zlib.compress(data, 1)
    with pytest.raises(NotImplementedError):
        rcmp.compress(12, 1, data, None)
    with pytest.raises(NotImplementedError):
        rcmp.de_compress(99, data, None)
    assert rcmp.import_checks({"compression_scheme": 0}) and rcmp.import_checks({"compression_scheme": 2})


def test_merge_interleaves_stream_mode_parts_by_frame_id(tmp_path):
    """Fixture G7: part files the reference's writer produced in mode='stream' (chunks of 5, 4, 1, 6 frames on 2 nodes: ids
    [0,1,2,5,6,9,10,11,12] and [3,4,7,8,13,14,15]).  merge_parts is a real k-way merge by frame id - the reference's own keeps the POPPED
    id as a part's next key and would emit 0,1,2,5,...: frame z of the merged file is the record with id z."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    g = load_npz("g7_stream.npz")
    by_id = {}
    for node in range(2):
        fn = "g7_stream.rc1_part%03d" % node
        shutil.copy(os.path.join(FILES, fn), tmp_path / fn)
        part = ReCoDeReader(str(tmp_path / fn), is_intermediate=True)
        part.open(print_header=False)
        ids = []
        while True:
            f = part.get_next_frame_raw()
            if f is None:
                break
            (fid, body), = f.items()
            ids.append(int(fid))
            by_id[int(fid)] = ({k: int(v) for k, v in body["metadata"].items()}, b"".join(body["data"].values()))
        assert ids == g["ids_part%d" % node].tolist()
        part.close()
    merge_parts(str(tmp_path), "g7_stream.rc1", 2)
    rd = ReCoDeReader(str(tmp_path / "g7_stream.rc1"), is_intermediate=False)
    rd.open(print_header=False)
    nz = g["frames"].shape[0]
    assert rd.get_shape()[0] == nz == len(by_id)
    for z in range(nz):
        (fid, body), = rd.get_next_frame_raw().items()
        assert fid == z
        assert {k: int(v) for k, v in body["metadata"].items()} == by_id[z][0]
        assert b"".join(body["data"].values()) == by_id[z][1], "frame %d" % z
    rd.close()


@pytest.mark.parametrize("host_decoded", [False, True])
def test_readahead_state_machine_without_a_gpu(monkeypatch, host_decoded):
    """ReCoDeReader._readahead_frame (the frame-at-a-time calls served from batches fetched ahead) with the batched call stubbed at
    get_frames_triplets: it starts with the third call in sequence, moves from batch to batch, drops its window on a jump, leaves empty
    frames and the frame counter / file position to the frame-at-a-time path, and turns itself off when a batch fails."""
    from pyrecode_amd.recode_reader import ReCoDeReader, _BatchOut
    g = load_npz("g3_l1z12.npz")
    frames, dark = g["frames"], g["dark"]
    want = np.where(frames > dark, frames - dark, 0).astype(np.uint16)
    nz = frames.shape[0]
    rd = ReCoDeReader(os.path.join(FILES, "g3_l1z12.rc1"), is_intermediate=False)
    rd.open(print_header=False)
    rd._RA_FRAMES = 3
    calls = []

    def fake(z0, n, out=None, coo=False):
        assert coo and out is rd._ra_buf
        calls.append((z0, n))
        rows, cols, vals, prefix = [], [], [], np.zeros(n + 1, np.uint64)
        for i in range(n):
            r, c = np.nonzero(want[z0 + i]) if z0 + i != 5 else (np.zeros(0, np.int64), np.zeros(0, np.int64))   # frame 5 pretends to be empty
            rows.append(r.astype(np.int32)); cols.append(c.astype(np.int32)); vals.append(want[z0 + i][r, c])
            prefix[i + 1] = prefix[i] + r.size
        rd.last_batch_path = "device"
        rd._current_frame_index = 99                     # (what a batched call leaves behind: the caller must not see it)
        rd._fp.seek(7, 0)
        return prefix, (np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))
    def fake_iter(z0, n, batch, coo=False):              # the host-decoded pipeline (zlib files, stock-encoder files), batch by batch
        assert coo
        for a in range(z0, z0 + n, batch):
            k = min(batch, z0 + n - a)
            yield (a,) + fake(a, k, out=rd._ra_buf, coo=True)
    monkeypatch.setattr(rd, "get_frames_triplets", fake)
    monkeypatch.setattr(rd, "_iter_frames_impl", fake_iter)          # (what the read-ahead drives: the public iterator's body)
    if not host_decoded:
        rd._header["compression_scheme"] = 1             # (a file of this library's own zstd streams ...
        rd._RA_PIPELINED = False                         #  ... through the other form: one synchronous batched call per window)
    rd._current_frame_index = 3
    rd._fp.seek(123, 0)
    assert rd._readahead_frame(0) is None and rd._readahead_frame(1) is None and calls == []
    m = rd._readahead_frame(2)
    assert calls == [(2, 3)] and np.array_equal(np.asarray(m.todense()), want[2]) and m.dtype == np.uint16
    assert rd._current_frame_index == 3 and rd._fp.tell() == 123               # put back
    assert np.array_equal(np.asarray(rd._readahead_frame(3).todense()), want[3]) and calls == [(2, 3)]
    assert np.array_equal(np.asarray(rd._readahead_frame(4).todense()), want[4]) and calls == [(2, 3)]
    assert rd._readahead_frame(5) is None and calls == [(2, 3), (5, 3)]          # the next batch; the empty frame is not served
    assert np.array_equal(np.asarray(rd._readahead_frame(6).todense()), want[6]) and rd.readahead_frames_served == 4
    assert rd._readahead_frame(1) is None and rd._ra is None                     # a jump back: window dropped, streak restarts
    assert rd._readahead_frame(2) is None
    assert np.array_equal(np.asarray(rd._readahead_frame(3).todense()), want[3]) and calls[-1] == (3, 3)
    assert rd._readahead_frame(7) is None and rd._ra is None and calls[-1] == (3, 3)     # a jump ahead, past the window's end: dropped as well
    assert rd._readahead_frame(6) is None                                          # (backwards again)

    def failing(z0, n, out=None, coo=False):
        raise ValueError("a damaged stream")
    def failing_iter(z0, n, batch, coo=False):
        raise ValueError("a damaged stream")
        yield
    monkeypatch.setattr(rd, "get_frames_triplets", failing)
    monkeypatch.setattr(rd, "_iter_frames_impl", failing_iter)
    rd._ra, rd._ra_last, rd._ra_streak = None, 0, 1
    assert rd._readahead_frame(1) is None and rd._ra_off is True
    assert rd._readahead_frame(2) is None
    rd.close()
    # the layout helper: triplet rows / the three COO arrays inside one buffer of `cap` entries
    buf = np.arange(10 * 7, dtype=np.uint8)
    r, c, v = _BatchOut.views(buf, 7, 5, True)
    assert r.dtype == np.int32 and r.size == 5 and r.ctypes.data == buf.ctypes.data
    assert c.ctypes.data == buf.ctypes.data + 4 * 7 and v.dtype == np.uint16 and v.ctypes.data == buf.ctypes.data + 8 * 7 and v.size == 5
    t = _BatchOut.views(np.zeros(24 * 7, np.uint8), 7, 5, False)
    assert t.shape == (5, 3) and t.dtype == np.uint64
    tr = np.array([[1, 2, 3], [4, 5, 70000]], np.uint64)
    rr, cc, vv = _BatchOut.from_triplets(tr, True)
    assert rr.tolist() == [1, 4] and cc.tolist() == [2, 5] and vv.dtype == np.uint16 and _BatchOut.from_triplets(tr, False) is tr


def test_batched_access_on_an_empty_part_file(tmp_path):
    """a part file that holds its header and nothing else (a writer whose block of the stack was empty): no frames, no batches, None"""
    from pyrecode_amd.recode_reader import ReCoDeReader
    src = open(os.path.join(FILES, "g3_l1z12.rc1_part000"), "rb").read()
    p = tmp_path / "empty.rc1_part000"
    p.write_bytes(src[:512])
    rd = ReCoDeReader(str(p), is_intermediate=True)
    rd.open(print_header=False)
    assert rd._batch_frames() == 0 and rd.part_frame_ids.size == 0
    assert list(rd.iter_frames_triplets(batch=4)) == [] and list(rd.iter_frames_coo(batch=4)) == []
    assert rd.get_next_frame() is None
    with pytest.raises(ValueError):
        rd.get_frames_triplets(0, 1)
    rd.close()


def test_frames_of_4_gib_and_more_are_refused_before_any_gpu_work():
    """A record may be as large as its raw frame and every size in a record is a u32 (structures.py:18-46): nx * ny * bytes_per_pixel >= 2^32 cannot
    be represented.  rc_ctx_create says so (RC_ERR_UNSUPPORTED -> NotImplementedError) instead of wrapping - argument checks come before the
    device is looked for, so this runs without a GPU."""
    import ctypes as C
    from pyrecode_amd import _lib
    L = _lib.lib()
    st = C.c_int(0)
    for nx, ny in ((50000, 50000), (65535, 32769), (46341, 46341)):          # uint16: 2 * nx * ny >= 2^32
        assert not L.rc_ctx_create(nx, ny, 16, 1, 1, 2, 1, 0, 4, C.byref(st))
        assert st.value == _lib.RC_ERR_UNSUPPORTED, (nx, ny, st.value)
        assert b"4 GiB" in L.rc_last_error()
    with pytest.raises(NotImplementedError):
        _lib.ReduceContext(50000, 50000, 16, 1, 1, 2, 1, 0, max_batch=1)
    assert not L.rc_ctx_create(70000, 70000, 16, 1, 1, 2, 1, 0, 4, C.byref(st)) and st.value == _lib.RC_ERR_BAD_ARG   # nx * ny itself >= 2^32
    assert not L.rc_ctx_create(46340, 46340, 16, 1, 1, 2, 1, 0, 4, C.byref(st)) and st.value == _lib.RC_ERR_DEVICE    # representable: only the GPU is missing here


def test_out_of_memory_has_a_status_and_an_exception_of_its_own():
    from pyrecode_amd import _lib
    assert _lib.lib().rc_strerror(_lib.RC_ERR_WORKSPACE) == b"out of device memory for the ctx's workspace"
    import pyrecode_amd._lib as m
    orig = m.last_error
    m.last_error = lambda: "hipMalloc"
    try:
        with pytest.raises(MemoryError):
            _lib.check(_lib.RC_ERR_WORKSPACE, "rc_ctx_create")
    finally:
        m.last_error = orig
