"""-m gpu: the reference-shaped Python API (ReCoDeWriter / ReCoDeReader / merge_parts / compressors / c_recode.Reader)
running on the HIP library, against files the reference itself wrote and against the oracle."""
import ctypes as C
import ctypes.util
import os
import shutil
import struct
import warnings

import numpy as np
import pytest

from conftest import GOLDEN, load_npz, synth_frames

pytestmark = pytest.mark.gpu
FILES = os.path.join(GOLDEN, "files")


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def _params(tmp_path, g, **over):
    from pyrecode_amd.params import InputParams
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    cfg.update(over)
    p = tmp_path / "params.txt"
    p.write_text("".join("%s = %d\n" % kv for kv in cfg.items()))
    ip = InputParams()
    ip.load(str(p))
    return ip, cfg


def _write_parts(tmp_path, base, dark, frames, nodes, g, batch_size=None, **over):
    from pyrecode_amd.recode_writer import ReCoDeWriter
    metrics = []
    for node in range(nodes):
        ip, cfg = _params(tmp_path, g, **over)
        w = ReCoDeWriter(base, dark_data=dark, output_directory=str(tmp_path), input_params=ip, mode="batch",
                         validation_frame_gap=-1, node_id=node, batch_size=batch_size)
        w.start()
        metrics.append(w.run(frames))
        w.close()
    return cfg, metrics


@pytest.mark.parametrize("tag,level,nodes", [("l1z12", 1, 3), ("l1z16", 1, 2), ("l1ro16", 1, 2), ("l3z", 3, 2)])
def test_writer_reproduces_reference_files_byte_for_byte(tag, level, nodes, tmp_path):
    """zlib / reduce-only configs: GPU reduce + pack, host zlib exactly as the reference calls it -> identical files."""
    from pyrecode_amd.recode_reader import merge_parts
    from pyrecode_amd import parallel
    g = load_npz("g3_%s.npz" % tag)
    base = "g3_" + tag
    cfg, metrics = _write_parts(tmp_path, base, g["dark"], g["frames"], nodes, g, batch_size=2)
    for node in range(nodes):
        fn = "%s.rc%d_part%03d" % (base, level, node)
        assert (tmp_path / fn).read_bytes() == open(os.path.join(FILES, fn), "rb").read(), fn
    assert sum(m["run_frames"] for m in metrics) == g["frames"].shape[0]
    for key in ("frame_thresholding_and_counting_time", "frame_binary_image_packing_time", "frame_time", "run_time"):
        assert key in metrics[0]
    merged = "%s.rc%d" % (base, level)
    want = open(os.path.join(FILES, merged), "rb").read()
    merge_parts(str(tmp_path), merged, nodes)
    assert (tmp_path / merged).read_bytes() == want
    os.remove(tmp_path / merged)
    recs = []
    for node in range(nodes):
        recs += parallel.read_part_records(os.path.join(tmp_path, "%s_part%03d" % (merged, node)))[1]
    parallel.merge_direct(str(tmp_path), merged, rank=0, world=1, records=recs)
    assert (tmp_path / merged).read_bytes() == want


@pytest.mark.parametrize("tag", ["l1z12", "l1z16", "l1ro16"])
def test_reader_decodes_reference_files(tag):
    from pyrecode_amd.recode_reader import ReCoDeReader
    g = load_npz("g3_%s.npz" % tag)
    frames, dark = g["frames"], g["dark"]
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    thr = (dark + np.uint16(cfg["calibration_threshold_epsilon"])).astype(np.uint16)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16)
    if g["decoded"].size:
        assert np.array_equal(want, g["decoded"])  # what the reference's own reader returned
    rd = ReCoDeReader(os.path.join(FILES, "g3_%s.rc1" % tag), is_intermediate=False)
    rd.open(print_header=False)
    for z in (3, 0, frames.shape[0] - 1, 1):  # random access
        f = rd.get_frame(z)
        assert list(f.keys()) == [z]
        coo = f[z]["data"]
        assert coo.dtype == np.uint16 and coo.shape == frames.shape[1:]
        assert np.array_equal(np.asarray(coo.todense()), want[z])
        assert np.all(np.diff(coo.row.astype(np.int64) * coo.shape[1] + coo.col) > 0)  # row-major order
    rd.close()
    part = ReCoDeReader(os.path.join(FILES, "g3_%s.rc1_part001" % tag), is_intermediate=True)
    part.open(print_header=False)
    n = 0
    while True:
        f = part.get_next_frame()
        if f is None:
            break
        (fid, body), = f.items()
        assert np.array_equal(np.asarray(body["data"].todense()), want[fid])
        n += 1
    assert n == part.get_header().as_dict()["nz"]
    part.close()


def test_l3_files_are_readable():
    from pyrecode_amd.recode_reader import ReCoDeReader
    g = load_npz("g3_l3z.npz")
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    thr = (g["dark"] + np.uint16(cfg["calibration_threshold_epsilon"])).astype(np.uint16)
    rd = ReCoDeReader(os.path.join(FILES, "g3_l3z.rc3"), is_intermediate=False)
    rd.open(print_header=False)
    for z in range(g["frames"].shape[0]):
        coo = rd.get_frame(z)[z]["data"]
        assert np.array_equal(np.asarray(coo.todense()) != 0, g["frames"][z] > thr)
    rd.close()


@pytest.mark.parametrize("depth,eps,nodes,scheme", [(12, 0, 3, 2), (16, 5, 2, 2), (12, 1, 2, 1), (16, 0, 3, 1), (12, 2, 2, 8), (16, 0, 3, 8)])
def test_device_codec_write_read_round_trip(depth, eps, nodes, scheme, tmp_path):
    """Config-2 shape of flow (L1 + LZ4 on device) at a test size: write parts, merge, read back, compare with the oracle's
    residual image; part files also decode through the oracle's LZ4 decoder (done in test_gpu_parity)."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, nz = 300, 420, 7
    dark, frames = synth_frames(77 + depth, nz, ny, nx, 0.02, depth)
    g = load_npz("g3_l1z12.npz")
    over = dict(num_rows=ny, num_cols=nx, num_frames=nz, num_threads=nodes, compression_scheme=scheme,
                source_bit_depth=depth, target_bit_depth=depth, calibration_threshold_epsilon=eps)
    _write_parts(tmp_path, "rt", dark, frames, nodes, g, batch_size=3, **over)
    merge_parts(str(tmp_path), "rt.rc1", nodes)
    thr = (dark + np.uint16(eps)).astype(np.uint16)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16)
    rd = ReCoDeReader(str(tmp_path / "rt.rc1"), is_intermediate=False)
    rd.open(print_header=False)
    assert rd.get_shape() == (nz, ny, nx)
    for z in range(nz):
        f = rd.get_next_frame()
        assert np.array_equal(np.asarray(f[z]["data"].todense()), want[z])
        md = f[z]["metadata"]
        assert md["bytes_in_packed_pixvals"] == (int((want[z] > 0).sum()) * depth + 7) // 8
    rd.close()


def test_validation_frames_and_dose_rate(tmp_path):
    from pyrecode_amd.recode_writer import ReCoDeWriter
    import scipy.ndimage as nd
    ny, nx, nz = 200, 256, 6
    dark, frames = synth_frames(5, nz, ny, nx, 0.01, 12)
    g = load_npz("g3_l1z12.npz")
    ip, _ = _params(tmp_path, g, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=1)
    w = ReCoDeWriter("val", dark_data=dark, output_directory=str(tmp_path), input_params=ip, validation_frame_gap=2, node_id=0)
    w.start()
    m = w.run(frames)
    w.close()
    raw = np.fromfile(tmp_path / "val_part000_validation_frames.bin", dtype=np.uint16).reshape(-1, ny, nx)
    assert np.array_equal(raw, frames[::2])
    want = []
    for z in range(0, nz, 2):
        roi = (frames[z] > dark)[36:164, 64:192]
        want.append(nd.label(roi, structure=nd.generate_binary_structure(2, 2))[1] / (128 * 128))
    assert m["run_dose_rates"] == want


def test_per_frame_seam_reduce_compress(tmp_path, orc):
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g3_l1z12.npz")
    ip, cfg = _params(tmp_path, g, num_threads=1)
    w = ReCoDeWriter("seam", dark_data=g["dark"], output_directory=str(tmp_path), input_params=ip, node_id=0)
    w.start()
    thr = orc.threshold(g["dark"], cfg["calibration_threshold_epsilon"])
    n, metrics, binary = w._reduce_compress(g["frames"][2], 41)
    want, _ = orc.l1_record(g["frames"][2], thr, 12, 41, mode=1)
    assert bytes(w._frame_buffer[:n]) == want
    assert binary.dtype == bool and np.array_equal(binary, g["frames"][2] > thr)
    w.close()


def test_compressor_seam_on_device(orc):
    from pyrecode_amd import recode_compressors as rcmp
    rng = np.random.default_rng(4)
    sparse = np.where(rng.random(70000) < 0.05, rng.integers(1, 256, 70000), 0).astype(np.uint8).tobytes()
    noise = rng.integers(0, 256, 5000).astype(np.uint8).tobytes()
    for data in (sparse, noise, b"", b"\x00" * 1, b"\x00" * 100000, b"abc" * 7, bytes(2048), bytes(2049)):
        c = rcmp.compress(2, 1, data, None)
        assert orc.lz4f_decode(c, len(data) + 8) == data
        assert rcmp.de_compress(2, c, None) == data
    assert len(rcmp.compress(2, 1, b"\x00" * 100000, None)) < 5000  # 196 blocks of 512 B x (4 + 19) + 11
    # streams written by stock liblz4 (linked 64 KiB blocks, what lz4.frame.compress produces) must decode too
    name = ctypes.util.find_library("lz4")
    if name:
        L = C.CDLL(name)
        L.LZ4F_compressFrameBound.restype = C.c_size_t
        L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
        L.LZ4F_compressFrame.restype = C.c_size_t
        L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        for data in (sparse * 3, noise):
            cap = L.LZ4F_compressFrameBound(len(data), None)
            dst = C.create_string_buffer(cap)
            n = L.LZ4F_compressFrame(dst, cap, data, len(data), None)
            assert rcmp.de_compress(2, dst.raw[:n], None) == data
    with pytest.raises(ValueError):
        rcmp.de_compress(2, b"\x04\x22\x4d\x18\x60\x40\x82" + b"\x05\x00\x00\x00" + b"\xff" * 5 + b"\x00" * 4, None)
    # zstd: encoded on the GPU, decoded by the stock library
    for data in (sparse, noise, b"", b"\x00", b"\x00" * 100000, b"abc" * 7, bytes(512), bytes(513)):
        c = rcmp.compress(1, 1, data, None)
        assert rcmp.de_compress(1, c, None) == data
    assert len(rcmp.compress(1, 1, b"\x00" * 100000, None)) < 1000
    # blosc1 (bit-shuffle + LZ4): encoded and decoded on the GPU; the from-spec decoder of the oracle must agree
    for data in (sparse, noise, b"", b"\x01", b"\x00" * 100000, b"abc" * 7, bytes(512), bytes(513), sparse[:700], noise[:64], noise[:63]):
        c = rcmp.compress(8, 1, data, None)
        assert orc.blosc1_decode(c) == data
        assert rcmp.de_compress(8, c, None) == data
    # a chunk laid out the way python-blosc would for a small array (byte shuffle, split streams, stored) built by hand
    import struct
    arr = rng.integers(0, 256, 128 * 8, dtype=np.uint8)           # 128 elements of typesize 8 -> blocksize/typesize >= 128: split
    sh = arr.reshape(128, 8).T.reshape(-1).tobytes()              # byte shuffle
    body = b"".join(struct.pack("<i", 128) + sh[j * 128:(j + 1) * 128] for j in range(8))
    chunk = bytes([2, 1, 0x21, 8]) + struct.pack("<iii", arr.size, arr.size, 16 + 4 + len(body)) + struct.pack("<i", 20) + body
    assert orc.blosc1_decode(chunk) == arr.tobytes()
    assert rcmp.de_compress(8, chunk, None) == arr.tobytes()


def test_c_recode_reader_shim(orc):
    from pyrecode_amd import c_recode
    rng = np.random.default_rng(8)
    ny, nx, d = 50, 70, 12
    binary = rng.random((ny, nx)) < 0.1
    vals = rng.integers(1, 4096, int(binary.sum())).astype(np.uint16)
    bitmap = np.packbits(binary.ravel(), bitorder="little").tobytes()
    r = c_recode.Reader()
    assert r.create_buffers(ny, nx, d) == 1
    packed = memoryview(bytearray((vals.size * d + 7) // 8))
    r.bit_pack_pixel_intensities(len(packed), vals.size, d, memoryview(bytearray(vals.tobytes())), packed)
    assert bytes(packed) == orc.bit_pack(vals, d).tobytes()
    buf = memoryview(bytearray(ny * nx * 3 * 8))  # sized like the reference does (recode_reader.py:111-115)
    n = r.get_frame_sparse(1, bitmap, bytes(packed), buf)
    assert n == vals.size
    trip = np.frombuffer(buf, np.uint64, count=n * 3).reshape(n, 3)
    assert np.array_equal(trip, orc.unpack_frame_sparse(nx, ny, d, np.frombuffer(bitmap, np.uint8), np.frombuffer(packed, np.uint8), 1))
    out = memoryview(bytearray(8 * vals.size))
    assert r.bit_unpack_pixel_intensities(vals.size, bytes(packed), out) == vals.size
    assert np.array_equal(np.frombuffer(out, np.uint64), vals.astype(np.uint64))
    assert r.count(bitmap) == vals.size


WORKER_2RANK = '''
import os, sys
sys.path.insert(0, %(repo)r)
import numpy as np
import torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)   # one GPU on the test box: ranks share it,
from pyrecode_amd import parallel                                             # the metadata exchange runs over gloo
from pyrecode_amd.params import InputParams
g = np.load(%(npz)r)
ip = InputParams(); ip.load(%(params)r)
m, nz = parallel.write_sharded("shard", g["frames"], g["dark"], %(out)r, ip, batch_size=3, device_id=0)
assert nz == g["frames"].shape[0], nz
dist.destroy_process_group()
'''


@pytest.mark.parametrize("nz,scheme", [(9, 2), (133, 1), (1, 2)])   # 133: 67 + 66 frames per rank (cfg-3 proportions); 1: rank 1 owns nothing
def test_two_rank_sharded_write_and_direct_merge(tmp_path, nz, scheme):
    """Multi-GPU driver end to end with 2 ranks (both on GPU 0 here; one rank per GPU on a real node): per-rank part files
    equal the single-writer part files, and the directly merged file equals merge_parts' result and decodes correctly."""
    import subprocess, sys
    from conftest import REPO
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx = 200, 300
    dark, frames = synth_frames(321, nz, ny, nx, 0.03, 12)
    g = load_npz("g3_l1z12.npz")
    ip, cfg = _params(tmp_path, g, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2, compression_scheme=scheme)
    np.savez(tmp_path / "in.npz", frames=frames, dark=dark)
    outdir = tmp_path / "dist"
    outdir.mkdir()
    script = tmp_path / "w.py"
    script.write_text(WORKER_2RANK % dict(repo=REPO, npz=str(tmp_path / "in.npz"), params=str(tmp_path / "params.txt"), out=str(outdir)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    # reference flow in this process: same two part files, file-based merge
    ref = tmp_path / "ref"
    ref.mkdir()
    _write_parts(ref, "shard", dark, frames, 2, g, batch_size=3, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2,
                 compression_scheme=scheme)
    merge_parts(str(ref), "shard.rc1", 2)
    for fn in ("shard.rc1_part000", "shard.rc1_part001", "shard.rc1"):
        assert (outdir / fn).read_bytes() == (ref / fn).read_bytes(), fn
    rd = ReCoDeReader(str(outdir / "shard.rc1"))
    rd.open(print_header=False)
    want = np.where(frames > dark, frames - dark, 0).astype(np.uint16)
    for z in sorted({0, nz // 2, nz - 1}):
        assert np.array_equal(np.asarray(rd.get_frame(z)[z]["data"].todense()), want[z])
    rd.close()


def test_l2_write_read_round_trip(tmp_path):
    """Level 2 through the reference-shaped API: ReCoDeWriter(reduction_level=2) -> merge -> ReCoDeReader returns the binary map
    plus 'summary_stats' (one value per 8-connected component, scipy label order)."""
    import scipy.ndimage as nd
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, nz = 120, 200, 5
    dark, frames = synth_frames(9, nz, ny, nx, 0.08, 12)
    g = load_npz("g3_l1z12.npz")
    over = dict(num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2, compression_scheme=2, reduction_level=2, l2_statistics=2)
    _write_parts(tmp_path, "l2", dark, frames, 2, g, batch_size=2, **over)
    merge_parts(str(tmp_path), "l2.rc2", 2)
    rd = ReCoDeReader(str(tmp_path / "l2.rc2"))
    rd.open(print_header=False)
    for z in range(nz):
        f = rd.get_frame(z)[z]
        binary = frames[z] > dark
        assert np.array_equal(np.asarray(f["data"].todense()) != 0, binary)
        labels, n = nd.label(binary, structure=np.ones((3, 3), int))
        want = (np.asarray(nd.sum(frames[z].astype(np.int64), labels, np.arange(1, n + 1)), np.int64) & 0xFFF).astype(np.uint16)
        assert np.array_equal(f["summary_stats"], want)   # 12-bit fields: a sum that does not fit wraps, as the reference's cast + pack would make it
        assert f["metadata"]["bytes_in_packed_summary_stats"] == (n * 12 + 7) // 8
    rd.close()


@pytest.mark.parametrize("kind", ["mrc", "seq", "mrc_short"])
def test_writer_file_mode_mrc_and_seq_sources(kind, tmp_path):
    """ReCoDeWriter(image_filename=<stack file>, dark_filename=<MRC file>).run() with no in-memory data
    (reference recode_writer.py:249-267, 327-348): frames and calibration frame come from pyrecode_amd.em_reader, the part
    files carry the 1024-byte source header that source_header_length announces, and the merged file decodes to the
    residual images.  "mrc_short": header and params claim 10 frames, the file holds 7 - the frame-by-frame
    fallback of the reference (:333-348) must load what is there."""
    from test_em_reader import write_mrc, write_seq
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    from pyrecode_amd.recode_writer import ReCoDeWriter
    from pyrecode_amd.misc import rc_cfg as rc
    ny, nx, nz, nodes, depth = 96, 160, 7, 2, 12
    dark, frames = synth_frames(5, nz, ny, nx, 0.03, depth)
    dark_path = str(tmp_path / "dark.mrc")
    write_mrc(dark_path, dark[None], 6)
    if kind.startswith("mrc"):
        src = str(tmp_path / "stack.mrc")
        src_hdr = write_mrc(src, frames, 6, nz_header=10 if kind == "mrc_short" else None)
        ftype = rc.FILE_TYPE_MRC
    else:
        src = str(tmp_path / "stack.seq")
        write_seq(src, frames.astype(np.int16), version=5)   # 16-bit sequences are int16 (values < 2^15 here)
        src_hdr = bytes(1024)
        ftype = rc.FILE_TYPE_SEQ
    g = load_npz("g3_l1z12.npz")
    over = dict(num_rows=ny, num_cols=nx, num_frames=10 if kind == "mrc_short" else nz, num_threads=nodes, compression_scheme=2, source_file_type=ftype,
                calibration_file_type=rc.FILE_TYPE_MRC, source_bit_depth=depth, target_bit_depth=depth)
    for node in range(nodes):
        ip, cfg = _params(tmp_path, g, **over)
        with pytest.warns(UserWarning) if kind == "seq" else _no_warning():
            w = ReCoDeWriter(src, dark_filename=dark_path, output_directory=str(tmp_path), input_params=ip, mode="batch",
                             node_id=node, batch_size=3)
            w.start()
            m = w.run()
            w.close()
        assert m["run_frames"] == ((5, 2) if kind == "mrc_short" else (4, 3))[node]
    part0 = (tmp_path / "stack.rc1_part000").read_bytes()
    assert part0[512:1536] == src_hdr
    merge_parts(str(tmp_path), "stack.rc1", nodes)
    want = np.where(frames > dark, frames - dark, 0).astype(np.uint16)
    rd = ReCoDeReader(str(tmp_path / "stack.rc1"), is_intermediate=False)
    rd.open(print_header=False)
    assert rd.get_shape() == (nz, ny, nx) and rd.get_source_header() == src_hdr
    for z in range(nz):
        assert np.array_equal(np.asarray(rd.get_frame(z)[z]["data"].todense()), want[z])
    rd.close()


class _no_warning:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


@pytest.mark.parametrize("scheme,clevel,mode,level,d", [(1, 1, 1, 1, 12), (1, 1, 1, 1, 16), (1, 0, 1, 1, 12), (2, 1, 1, 1, 12), (2, 1, 1, 3, 12),
                                                       (1, 1, 1, 3, 12), (0, 1, 0, 1, 12), (0, 1, 1, 1, 12)])
def test_batched_reader_equals_per_frame_reader(tmp_path, scheme, clevel, mode, level, d):
    """ReCoDeReader.get_frames_triplets (rc_expand_frames: every frame's two streams decoded and expanded on the GPU in one
    call - device zstd / LZ4 decoders for this library's own frames, the per-frame stock-decoder path for zlib) must return
    exactly what the reference-shaped per-frame reader returns, frame by frame, in row-major order."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, nz = 150, 260, 7
    dark, frames = synth_frames(50 + scheme + d, nz, ny, nx, 0.04, d)
    frames[3] = dark // 2                       # an empty frame in the middle
    g = load_npz("g3_l1z12.npz")
    over = dict(num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2, compression_scheme=scheme, compression_level=clevel,
                rc_operation_mode=mode, reduction_level=level, source_bit_depth=d, target_bit_depth=d)
    _write_parts(tmp_path, "bt", dark, frames, 2, g, batch_size=3, **over)
    base = "bt.rc%d" % level
    merge_parts(str(tmp_path), base, 2)
    rd = ReCoDeReader(str(tmp_path / base))
    rd.open(print_header=False)
    prefix, trip = rd.get_frames_triplets(1, nz - 1)
    assert prefix[0] == 0 and prefix[-1] == trip.shape[0]
    assert rd.last_batch_path == ("host-decode + device-expand" if (mode == 1 and scheme == 0) else "device")   # zlib: the host library on the pool
    want = np.where(frames > dark, frames - dark if level == 1 else 1, 0)
    for i, z in enumerate(range(1, nz)):
        t = trip[int(prefix[i]):int(prefix[i + 1])]
        rows, cols = np.nonzero(want[z])
        assert np.array_equal(t[:, 0], rows.astype(np.uint64)) and np.array_equal(t[:, 1], cols.astype(np.uint64)), z
        assert np.array_equal(t[:, 2], want[z][rows, cols].astype(np.uint64)), z
    fr = rd.get_frames(0, 2)
    assert np.array_equal(np.asarray(fr[1]["data"].todense()), want[1].astype(np.uint16))
    # the streaming form (two batches in flight) walks the whole file in batches of 3 and of 2 and yields the same triplets
    for batch in (3, 2):
        seen = 0
        for a, pfx, t3 in rd.iter_frames_triplets(0, nz, batch=batch):
            k = len(pfx) - 1
            assert a == seen and k == min(batch, nz - a)
            for i in range(k):
                t = t3[int(pfx[i]):int(pfx[i + 1])]
                rows, cols = np.nonzero(want[a + i])
                assert np.array_equal(t[:, 0], rows.astype(np.uint64)) and np.array_equal(t[:, 1], cols.astype(np.uint64)), a + i
                assert np.array_equal(t[:, 2], want[a + i][rows, cols].astype(np.uint64)), a + i
            seen += k
        assert seen == nz
    it = rd.iter_frames_triplets(0, nz, batch=2)     # a consumer that stops after the first batch: the queued one is waited for
    next(it)
    it.close()
    prefix2, trip2 = rd.get_frames_triplets(1, nz - 1)
    assert np.array_equal(prefix2, prefix) and np.array_equal(trip2, trip)
    rd.close()


def test_reference_minimal_read_write_test_at_its_own_size(tmp_path, orc):
    """BASELINE configs[0] literally (reference tests/minimal_read_write_test.py:16-43,82-119): 9 x 512x512 uint16 frames of
    randint(0, 4096) - 3500 clipped at 0, zero dark frame, the reference's own parameter file (L1, zlib-1, 12 bit, 3 nodes,
    validation_frame_gap 2), one writer per node -> part files -> merge_parts -> sequential and random-access read.  Every
    record is compared byte for byte with the oracle's record (zlib through the same stdlib call the reference makes)."""
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    from pyrecode_amd.recode_writer import ReCoDeWriter
    shape = (9, 512, 512)
    rng = np.random.RandomState(20261004)
    data = rng.randint(0, high=4096, size=shape) - 3500
    data[data < 0] = 0
    data = data.astype(np.uint16)
    calib = np.zeros(shape[1:], np.uint16)
    ip = InputParams()
    ip.load(os.path.join(FILES, "recode_params_minimal_read_write_test.txt"))
    ip.nx, ip.ny, ip.nz = shape[1], shape[2], shape[0]      # as the reference's test does (:36-40)
    ip.source_data_type = 0
    ip.target_data_type = 0
    assert (ip.num_threads, ip.compression_scheme, ip.source_bit_depth, ip.num_frames) == (3, 0, 12, 9)
    for node in range(3):
        w = ReCoDeWriter("test_data", dark_data=calib, output_directory=str(tmp_path), input_params=ip, mode="batch",
                         validation_frame_gap=2, log_filename=str(tmp_path / "recode.log"), run_name="minimal_read_write_test",
                         verbosity=0, use_c=False, node_id=node)
        w.start()
        m = w.run(data)
        w.close()
        assert m["run_frames"] == 3
        import scipy.ndimage as nd
        want_rates = []
        for z in range(3 * node, 3 * node + 3):
            if z % 2 == 0:   # central 128 x 128 of the binary map, 8-connected components / ROI pixels (recode_writer.py:161,402-415)
                roi = data[z, 192:320, 192:320] > 0
                want_rates.append(nd.label(roi, structure=nd.generate_binary_structure(2, 2))[1] / (128 * 128))
        assert list(m["run_dose_rates"]) == want_rates
    thr = orc.threshold(calib, 0)
    for node in range(3):
        blob = (tmp_path / ("test_data.rc1_part%03d" % node)).read_bytes()
        want = b"".join(orc.l1_record(data[z], thr, 12, z, mode=1)[0] for z in range(3 * node, 3 * node + 3))
        assert blob[512:] == want, "part %d" % node
    # the reference's intermediate-file read loop (:82-93), with an exact comparison instead of its sum test
    rd = ReCoDeReader(str(tmp_path / "test_data.rc1_part000"), is_intermediate=True)
    rd.open(print_header=False)
    hdr = rd.get_header().as_dict()
    assert hdr["nz"] == 3
    for _ in range(hdr["nz"]):
        f = rd.get_next_frame()
        fid = list(f.keys())[0]
        assert np.sum(data[fid] - f[fid]["data"].todense()) == 0                # the reference's own check
        assert np.array_equal(np.asarray(f[fid]["data"].todense()), data[fid])
    rd.close()
    merge_parts(str(tmp_path), "test_data.rc1", 3)
    merged = (tmp_path / "test_data.rc1").read_bytes()
    recs = [orc.l1_record(data[z], thr, 12, z, mode=1)[0] for z in range(9)]
    md = b"".join(r[4:16] for r in recs)
    assert merged[512:] == md + b"".join(r[16:] for r in recs)
    rd = ReCoDeReader(str(tmp_path / "test_data.rc1"), is_intermediate=False)
    rd.open(print_header=False)
    for i in range(9):
        f = rd.get_next_frame()
        assert np.array_equal(np.asarray(f[i]["data"].todense()), data[i])
    for i in (7, 0, 4):
        assert np.array_equal(np.asarray(rd.get_frame(i)[i]["data"].todense()), data[i])
    rd.close()
    # validation frames: raw frames whose absolute index is a multiple of the gap (recode_writer.py:402-415), one dose rate each
    for node, ids in ((0, [0, 2]), (1, [4]), (2, [6, 8])):
        v = np.fromfile(tmp_path / ("test_data_part%03d_validation_frames.bin" % node), np.uint16).reshape(-1, 512, 512)
        assert v.shape[0] == len(ids) and all(np.array_equal(v[k], data[z]) for k, z in enumerate(ids))


def test_validation_frames_ride_the_streaming_path(tmp_path, orc):
    """validation_frame_gap > 0 (reference recode_writer.py:402-415) no longer leaves the streaming writer: raw frames go to the
    validation file, the ROI's component count comes from the device (rc_pipe_validation) - compared with scipy.ndimage.label
    on clustered frames, a frame whose ROI holds ONE long spiral (many propagation sweeps), an empty and a full ROI, with
    batches that start at ids which are not multiples of the gap; the records are the ones written without validation."""
    import scipy.ndimage as nd
    from pyrecode_amd import synth
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_writer import ReCoDeWriter
    ny, nx, nz, gap = 200, 300, 11, 3
    dark = synth.dark_frame(5, ny * nx).reshape(ny, nx)
    data = synth.frames_clustered(5, 0, nz, nx, ny, 30000, dark).reshape(nz, ny, nx)
    y0, x0 = (ny - 128) // 2, (nx - 128) // 2
    roi = np.zeros((128, 128), bool)                        # frame 3: ONE serpentine component, ~8000 pixels end to end
    roi[0::2, :] = True
    for r in range(1, 127, 2):
        roi[r, 127 if (r // 2) % 2 == 0 else 0] = True
    data[3] = dark // 2
    data[3, y0:y0 + 128, x0:x0 + 128][roi] = dark[y0:y0 + 128, x0:x0 + 128][roi] + 9
    data[6] = dark // 2                                      # empty ROI
    data[9] = (dark + 1).astype(np.uint16)                   # every pixel set: one component
    ip = InputParams()
    ip._param_map.update(dict(reduction_level=1, rc_operation_mode=1, calibration_threshold_epsilon=0, target_bit_depth=12,
                              source_bit_depth=12, num_cols=nx, num_rows=ny, num_frames=nz, frame_offset=0, num_calibration_frames=1,
                              calibration_frame_offset=0, keep_part_files=1, num_threads=1, l2_statistics=0, l4_centroiding=0,
                              compression_scheme=2, compression_level=1, source_file_type=0, source_header_length=0,
                              keep_calibration_data=0, calibration_file_type=0, source_data_type=0, target_data_type=0))
    parts = {}
    for tag, g in (("with", gap), ("without", -1)):
        out = tmp_path / tag
        out.mkdir()
        w = ReCoDeWriter("stack.bin", dark_data=dark, output_directory=str(out), input_params=ip, mode="batch", validation_frame_gap=g,
                         node_id=0, batch_size=4)
        w.start()
        m = w.run(data)
        w.close()
        parts[tag] = (out / "stack.rc1_part000").read_bytes()
        if g > 0:
            ids = [z for z in range(nz) if z % gap == 0]
            want = []
            for z in ids:
                r = data[z, y0:y0 + 128, x0:x0 + 128] > dark[y0:y0 + 128, x0:x0 + 128]
                want.append(nd.label(r, structure=nd.generate_binary_structure(2, 2))[1] / (128 * 128))
            assert list(m["run_dose_rates"]) == want
            assert want[1] == 1 / (128 * 128) and want[2] == 0 and want[3] == 1 / (128 * 128)
            v = np.fromfile(out / "stack_part000_validation_frames.bin", np.uint16).reshape(-1, ny, nx)
            assert v.shape[0] == len(ids) and all(np.array_equal(v[k], data[z]) for k, z in enumerate(ids))
        else:
            assert "run_dose_rates" not in m
    assert parts["with"] == parts["without"]


def test_failed_run_leaves_the_validation_file_where_the_part_file_ends(tmp_path):
    """The streaming writer queues a run's validation frames for the side file before the batches are processed.  A run that fails
    half way - here a frame whose record would exceed its raw size, the reference's ValueError (recode_writer.py:565-566) - must not
    leave the side file ahead of the part file: it is cut back to the validation frames of the batches whose records were appended."""
    from pyrecode_amd import synth
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_writer import ReCoDeWriter
    ny, nx, nz, gap = 120, 250, 11, 3
    dark = synth.dark_frame(9, ny * nx).reshape(ny, nx)
    data = synth.frames(9, 0, nz, ny * nx, 20000, dark.reshape(-1)).reshape(nz, ny, nx)
    data[9] = 40000                                          # every pixel set at 16 bits: bitmap + 2 N bytes > the raw frame
    ip = InputParams()
    ip._param_map.update(dict(reduction_level=1, rc_operation_mode=1, calibration_threshold_epsilon=0, target_bit_depth=16,
                              source_bit_depth=16, num_cols=nx, num_rows=ny, num_frames=nz, frame_offset=0, num_calibration_frames=1,
                              calibration_frame_offset=0, keep_part_files=1, num_threads=1, l2_statistics=0, l4_centroiding=0,
                              compression_scheme=2, compression_level=1, source_file_type=0, source_header_length=0,
                              keep_calibration_data=0, calibration_file_type=0, source_data_type=0, target_data_type=0))
    w = ReCoDeWriter("stack.bin", dark_data=dark, output_directory=str(tmp_path), input_params=ip, mode="batch", validation_frame_gap=gap,
                     node_id=0, batch_size=4)
    w.start()
    with pytest.raises(ValueError, match="Buffer size smaller than compressed data size"):
        w.run(data)
    w.close()
    # batches [0..3] and [4..7] made it: eight records, validation frames 0, 3, 6 - frame 9's batch did not
    v = np.fromfile(tmp_path / "stack_part000_validation_frames.bin", np.uint16).reshape(-1, ny, nx)
    assert v.shape[0] == 3 and all(np.array_equal(v[k], data[z]) for k, z in enumerate((0, 3, 6)))
    from pyrecode_amd import parallel
    ids = [r[0] for r in parallel.read_part_records(str(tmp_path / "stack.rc1_part000"))[1]]
    assert ids == list(range(8))


def _stock_encoders():
    """{scheme: bytes -> stock-encoded frame} through the system libraries (ctypes), as the reference's packages would write them:
    libzstd level 1 (4-stream literals, real offsets, 128 KiB blocks), liblz4 frames with default preferences (linked 64 KiB blocks)."""
    enc = {}
    name = ctypes.util.find_library("zstd")
    if name:
        z = C.CDLL(name)
        z.ZSTD_compress.restype = C.c_size_t
        z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]

        def zenc(b):
            dst = C.create_string_buffer(z.ZSTD_compressBound(len(b)) + 64)
            n = z.ZSTD_compress(dst, len(dst), b, len(b), 1)     # (before dst.raw is taken: that is a copy)
            return dst.raw[:n]
        enc[1] = zenc
    name = ctypes.util.find_library("lz4")
    if name:
        lz = C.CDLL(name)
        lz.LZ4F_compressFrameBound.restype = C.c_size_t
        lz.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
        lz.LZ4F_compressFrame.restype = C.c_size_t
        lz.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]

        def lenc(b):
            dst = C.create_string_buffer(lz.LZ4F_compressFrameBound(len(b), None) + 64)
            n = lz.LZ4F_compressFrame(dst, len(dst), b, len(b), None)
            return dst.raw[:n]
        enc[2] = lenc
    return enc


@pytest.mark.parametrize("scheme", [1, 2])
def test_files_a_stock_encoder_wrote_take_the_batched_path(scheme, tmp_path, orc):
    """A merged file whose streams the STOCK libraries encoded (what the reference's writer produces with zstandard / lz4.frame):
    the device decoders refuse them, and the batch is decoded by the stock library on a thread pool and expanded by ONE device
    call - not frame by frame.  Sequential batches, the streaming iterator and single frames all give the oracle's triplets."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    enc = _stock_encoders().get(scheme)
    if enc is None:
        pytest.skip("no system library for scheme %d" % scheme)
    ny, nx, d, nz = 600, 1100, 12, 7        # the binary map (82 500 bytes) spans two of liblz4's linked 64 KiB blocks
    dark, frames = synth_frames(31, nz, ny, nx, 0.02, d)
    g = load_npz("g3_l1z12.npz")
    _write_parts(tmp_path, "own", dark, frames, 1, g, batch_size=4, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=1,
                 compression_scheme=scheme, calibration_threshold_epsilon=0)
    merge_parts(str(tmp_path), "own.rc1", 1)
    own = (tmp_path / "own.rc1").read_bytes()
    thr = orc.threshold(dark, 0)
    md, blobs = [], []
    for z in range(nz):
        bitmap, packed, nnz = orc.reduce_frame_l1(frames[z].ravel(), thr.ravel(), d)
        cb, cp = enc(bitmap.tobytes()), enc(packed.tobytes())
        md.append(struct.pack("<III", len(cb), len(cp), packed.size))
        blobs.append(cb + cp)
    foreign = tmp_path / "foreign.rc1"
    foreign.write_bytes(own[:512] + b"".join(md) + b"".join(blobs))
    want = [orc.unpack_frame_sparse(nx, ny, d, *orc.reduce_frame_l1(frames[z].ravel(), thr.ravel(), d)[:2], 1) for z in range(nz)]
    rd = ReCoDeReader(str(foreign), is_intermediate=False)
    rd.open(print_header=False)
    prefix, trip = rd.get_frames_triplets(0, nz)
    assert rd.last_batch_path == "host-decode + device-expand"
    for z in range(nz):
        assert np.array_equal(trip[int(prefix[z]):int(prefix[z + 1])], want[z]), "frame %d" % z
    seen = 0
    for a, pre, tr in rd.iter_frames_triplets(0, nz, batch=3):
        assert rd.last_batch_path == "host-decode + device-expand"
        for i in range(len(pre) - 1):
            assert np.array_equal(tr[int(pre[i]):int(pre[i + 1])], want[a + i])
            seen += 1
    assert seen == nz
    f = rd.get_frame(5)[5]["data"]                                   # the reference's frame-at-a-time API on the same file
    assert np.array_equal(np.asarray(f.todense()), np.where(frames[5] > thr, frames[5] - thr, 0))
    rd.close()
    # a fresh reader finds out INSIDE the streaming iterator (its first two batches are refused by the device decoders) and moves the
    # rest of the file to the host-decoded pipeline, which decodes one batch ahead of the device
    rd = ReCoDeReader(str(foreign), is_intermediate=False)
    rd.open(print_header=False)
    seen, paths = 0, []
    for a, pre, tr in rd.iter_frames_triplets(0, nz, batch=2):
        paths.append(rd.last_batch_path)
        assert a == seen
        for i in range(len(pre) - 1):
            assert np.array_equal(tr[int(pre[i]):int(pre[i + 1])], want[a + i])
            seen += 1
    assert seen == nz and set(paths) == {"host-decode + device-expand"}
    it = rd.iter_frames_triplets(1, nz - 1, batch=2)                 # a consumer that stops early, with a decode running ahead
    a, pre, tr = next(it)
    assert a == 1 and np.array_equal(tr[:int(pre[1])], want[1])
    p2, t2 = rd.get_frames_triplets(0, nz)                           # a synchronous call BETWEEN the iterator's steps has buffers of its own
    for z in range(nz):
        assert np.array_equal(t2[int(p2[z]):int(p2[z + 1])], want[z])
    a, pre, tr = next(it)
    assert a == 3 and np.array_equal(tr[:int(pre[1])], want[3]) and np.array_equal(tr[int(pre[1]):int(pre[2])], want[4])
    it.close()
    prefix, trip = rd.get_frames_triplets(2, 3)
    for i in range(3):
        assert np.array_equal(trip[int(prefix[i]):int(prefix[i + 1])], want[2 + i])
    rd.close()
    # a damaged stream in such a file: the batches in front of it arrive, its own batch raises (the stock decoder is the judge), and the
    # reader still serves the intact frames afterwards
    raw = bytearray(foreign.read_bytes())
    data0 = 512 + nz * 12
    at = data0 + sum(len(b) for b in blobs[:4]) + len(blobs[4]) // 4        # inside frame 4's binary-map stream
    for j in range(8):
        raw[at + j] ^= 0xFF
    damaged = tmp_path / "damaged.rc1"
    damaged.write_bytes(bytes(raw))
    rd = ReCoDeReader(str(damaged), is_intermediate=False)
    rd.open(print_header=False)
    got = []
    with pytest.raises(Exception):
        for a, pre, tr in rd.iter_frames_triplets(0, nz, batch=2):
            got.append(a)
            for i in range(len(pre) - 1):
                assert np.array_equal(tr[int(pre[i]):int(pre[i + 1])], want[a + i])
    assert got == [0, 2]
    prefix, trip = rd.get_frames_triplets(5, 2)
    for i in range(2):
        assert np.array_equal(trip[int(prefix[i]):int(prefix[i + 1])], want[5 + i])
    rd.close()


@pytest.mark.parametrize("tag", ["l1z12", "l1z16"])
def test_reference_zlib_files_take_the_batched_path(tag):
    """Files the REFERENCE wrote with its default codec (zlib, host-only) are read in batches too: the stock decoder on the thread pool,
    one device expand - against what the reference's own reader returned for them (fixture G3)."""
    from pyrecode_amd.recode_reader import ReCoDeReader
    g = load_npz("g3_%s.npz" % tag)
    frames, dark = g["frames"], g["dark"]
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    thr = (dark + np.uint16(cfg["calibration_threshold_epsilon"])).astype(np.uint16)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16)
    rd = ReCoDeReader(os.path.join(FILES, "g3_%s.rc1" % tag), is_intermediate=False)
    rd.open(print_header=False)
    nz, (ny, nx) = frames.shape[0], frames.shape[1:]
    prefix, trip = rd.get_frames_triplets(0, nz)
    assert rd.last_batch_path == "host-decode + device-expand"
    seen = 0
    for a, pre, tr in rd.iter_frames_triplets(0, nz, batch=2):
        assert np.array_equal(tr, trip[int(prefix[a]):int(prefix[a + len(pre) - 1])])
        seen += len(pre) - 1
    assert seen == nz
    for z in range(nz):
        t = trip[int(prefix[z]):int(prefix[z + 1])]
        dense = np.zeros((ny, nx), np.uint16)
        dense[t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.uint16)
        assert np.array_equal(dense, want[z]), "frame %d" % z
    rd.close()


@pytest.mark.parametrize("tag,level,nodes", [("l1z12", 1, 3), ("l1ro16", 1, 2), ("l3z", 3, 2)])
def test_part_files_the_reference_wrote_stream_through_the_batched_reader(tag, level, nodes):
    """The reference's second test sums the frames of a PART file one get_next_frame at a time (tests/recode_v1_read_test.py:9-21).
    Part files have no metadata table; the batched readers index their records once and then take them like a merged file's frames:
    every part file of fixture G3 (zlib = host decode + device expand, reduce-only and level 3 = device) against the frames the
    reference's own reader returned, the sum the reference test computes, and the sequential reader on the same file afterwards."""
    from pyrecode_amd.recode_reader import ReCoDeReader
    g = load_npz("g3_%s.npz" % tag)
    frames, dark = g["frames"], g["dark"]
    cfg = dict(zip(g["cfg_keys"].tolist(), (int(v) for v in g["cfg_vals"])))
    thr = (dark + np.uint16(cfg["calibration_threshold_epsilon"])).astype(np.uint16)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16) if level == 1 else (frames > thr).astype(np.uint16)
    nz, (ny, nx) = frames.shape[0], frames.shape[1:]
    seen = []
    for node in range(nodes):
        rd = ReCoDeReader(os.path.join(FILES, "g3_%s.rc%d_part%03d" % (tag, level, node)), is_intermediate=True)
        rd.open(print_header=False)
        first = rd.get_next_frame()                                  # the sequential cursor sits behind the first record ...
        summed = np.zeros((ny, nx), np.int64)
        for a, pre, tr in rd.iter_frames_triplets(batch=2):
            for i in range(len(pre) - 1):
                fid = int(rd.part_frame_ids[a + i])
                t = tr[int(pre[i]):int(pre[i + 1])]
                dense = np.zeros((ny, nx), np.uint16)
                dense[t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.uint16) if level == 1 else 1
                assert np.array_equal(dense, want[fid]), "part %d, frame id %d" % (node, fid)
                summed += dense
                seen.append(fid)
        ids = rd.part_frame_ids.tolist()
        assert np.array_equal(summed, want[ids].astype(np.int64).sum(axis=0))
        got = rd.get_frames(0, len(ids))                             # keyed by frame id, like get_next_frame
        assert sorted(got) == ids
        for fid in ids:
            assert np.array_equal(np.asarray(got[fid]["data"].todense()) != 0, want[fid] != 0)
        prefix, trip = rd.get_frames_triplets(len(ids) - 1, 1)       # positional access to the last record
        assert int(prefix[1]) == int((want[ids[-1]] != 0).sum())
        second = rd.get_next_frame()                                 # ... and is still there
        if len(ids) > 1:
            assert list(first) == [ids[0]] and list(second) == [ids[1]]
        with pytest.raises(ValueError):
            rd.get_frame(0)                                          # (the reference's rule for its frame-at-a-time call stays)
        rd.close()
    assert sorted(seen) == list(range(nz))


@pytest.mark.parametrize("batch_size", [None, 2])
def test_stream_mode_writer_reproduces_the_references_part_files(tmp_path, batch_size):
    """mode='stream' (reference recode_writer.py:193-194,311-322,422-423): the writer is handed chunk after chunk, every chunk is split
    over the nodes by the contiguous-block rule and the frame ids run on from chunk to chunk - a part file's ids are increasing but not
    contiguous.  Fixture G7: the reference's own writer fed chunks of 5, 4, 1 and 6 frames on 2 nodes (a node that gets NOTHING of the
    one-frame chunk included).  The part files are the reference's byte for byte; merge_parts then interleaves them by frame id (the
    reference's own merge misorders such parts) and both readers return every frame in its place."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g7_stream.npz")
    dark, frames, chunks, nodes = g["dark"], g["frames"], g["chunks"].tolist(), int(g["n_nodes"])
    base = "g7_stream"
    for node in range(nodes):
        ip, cfg = _params(tmp_path, g)
        w = ReCoDeWriter(base, dark_data=dark, output_directory=str(tmp_path), input_params=ip, mode="stream", validation_frame_gap=-1,
                         node_id=node, run_name=base, batch_size=batch_size)
        w.start()
        at, took = 0, 0
        for c in chunks:
            took += w.run(frames[at:at + c])["run_frames"]
            at += c
        w.close()
        fn = "%s.rc1_part%03d" % (base, node)
        assert (tmp_path / fn).read_bytes() == open(os.path.join(FILES, fn), "rb").read(), fn
        assert took == len(g["ids_part%d" % node])
    thr = (dark + np.uint16(cfg["calibration_threshold_epsilon"])).astype(np.uint16)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16)
    nz = frames.shape[0]
    # the batched reader on a part file with gaps in its ids
    part = ReCoDeReader(str(tmp_path / (base + ".rc1_part000")), is_intermediate=True)
    part.open(print_header=False)
    got_ids = []
    for a, pre, tr in part.iter_frames_triplets(batch=4):
        for i in range(len(pre) - 1):
            fid = int(part.part_frame_ids[a + i])
            t = tr[int(pre[i]):int(pre[i + 1])]
            dense = np.zeros(want.shape[1:], np.uint16)
            dense[t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.uint16)
            assert np.array_equal(dense, want[fid])
            got_ids.append(fid)
    assert got_ids == g["ids_part0"].tolist()
    part.close()
    merge_parts(str(tmp_path), base + ".rc1", nodes)
    rd = ReCoDeReader(str(tmp_path / (base + ".rc1")), is_intermediate=False)
    rd.open(print_header=False)
    assert rd.get_shape()[0] == nz
    for z in (0, 3, 5, 9, 12, 13, nz - 1):
        assert np.array_equal(np.asarray(rd.get_frame(z)[z]["data"].todense()).astype(np.uint16), want[z]), "frame %d" % z
    prefix, trip = rd.get_frames_triplets(0, nz)
    for z in range(nz):
        t = trip[int(prefix[z]):int(prefix[z + 1])]
        dense = np.zeros(want.shape[1:], np.uint16)
        dense[t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.uint16)
        assert np.array_equal(dense, want[z]), "frame %d" % z
    rd.close()


@pytest.mark.parametrize("batch_size", [None, 3])
def test_validation_frames_and_dose_rates_equal_the_references(tmp_path, batch_size):
    """Fixture G8: the reference's writer with validation_frame_gap = 3 on 10 frames of clustered events (150 x 170: the 128 x 128 ROI
    is a true sub-window), 2 nodes.  Here the ROI's components are counted on the device (k_roi_components) while the batch streams
    through: part files, validation side files and every dose rate equal the reference's own (recode_writer.py:400-415)."""
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g8_valid.npz")
    dark, frames, nodes, gap = g["dark"], g["frames"], int(g["n_nodes"]), int(g["gap"])
    base = "g8_valid"
    for node in range(nodes):
        ip, cfg = _params(tmp_path, g)
        w = ReCoDeWriter(base, dark_data=dark, output_directory=str(tmp_path), input_params=ip, mode="batch", validation_frame_gap=gap,
                         node_id=node, batch_size=batch_size)
        w.start()
        m = w.run(frames)
        w.close()
        fn = "%s.rc1_part%03d" % (base, node)
        assert (tmp_path / fn).read_bytes() == open(os.path.join(FILES, fn), "rb").read(), fn
        side = (tmp_path / ("%s_part%03d_validation_frames.bin" % (base, node))).read_bytes()
        assert side == g["validation_part%d" % node].tobytes()
        assert list(m["run_dose_rates"]) == g["rates_part%d" % node].tolist()


@pytest.mark.parametrize("tag,level,has_merged", [("l1bz2", 1, False), ("l1lzma", 1, False), ("l1z12_lvl9", 1, True), ("l3ro", 3, True)])
def test_other_host_schemes_reproduce_the_references_files(tag, level, has_merged, tmp_path):
    """Fixture G9: bz2 and lzma (recode_compressors.py:97-101), zlib at level 9, level 3 without compression - the reference's part files
    byte for byte (and its merged file where its own reader can open the parts: it has no import entry for bz2 / lzma, so those stop at
    the part files).  Every merged file then reads back to the expected frames through the batched reader."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    g = load_npz("g9_%s.npz" % tag)
    nodes, base = int(g["n_nodes"]), "g9_" + tag
    dark, frames = g["dark"], g["frames"]
    cfg, _ = _write_parts(tmp_path, base, dark, frames, nodes, g, batch_size=2)
    for node in range(nodes):
        fn = "%s.rc%d_part%03d" % (base, level, node)
        assert (tmp_path / fn).read_bytes() == open(os.path.join(FILES, fn), "rb").read(), fn
    merged = "%s.rc%d" % (base, level)
    merge_parts(str(tmp_path), merged, nodes)
    if has_merged:
        assert (tmp_path / merged).read_bytes() == open(os.path.join(FILES, merged), "rb").read()
    thr = (dark + np.uint16(cfg["calibration_threshold_epsilon"])).astype(np.uint16)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16) if level == 1 else (frames > thr).astype(np.uint16)
    if level == 1 and has_merged:
        assert np.array_equal(g["decoded"], want)                      # (what the reference's own reader returned)
    rd = ReCoDeReader(str(tmp_path / merged), is_intermediate=False)
    rd.open(print_header=False)
    prefix, trip = rd.get_frames_triplets(0, frames.shape[0])
    for z in range(frames.shape[0]):
        t = trip[int(prefix[z]):int(prefix[z + 1])]
        dense = np.zeros(want.shape[1:], np.uint16)
        dense[t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.uint16) if level == 1 else 1
        assert np.array_equal(dense, want[z]), "frame %d" % z
    rd.close()


def _coo_equal(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.row, b.row) and np.array_equal(a.col, b.col) and np.array_equal(a.data, b.data)


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_frame_at_a_time_calls_read_ahead_and_return_the_same_frames(scheme, tmp_path, orc):
    """The reference's calls - get_frame(z) in a loop, get_next_frame() - are served out of the batched reader once they turn out to
    be sequential (ReCoDeReader._readahead_frame).  Same dictionaries, same COO arrays, same file positions as a reader that goes
    frame by frame (_ra_off), on a merged file with an EMPTY frame in the middle (whose conventions stay with the frame-at-a-time
    path), on its part files, with jumps, repeats and a switch between the two calls."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, d, nz = 96, 200, 12, 23
    dark, frames = synth_frames(77, nz, ny, nx, 0.03, d)
    frames[9] = 0                                                        # an empty frame
    g = load_npz("g3_l1z12.npz")
    _write_parts(tmp_path, "ra", dark, frames, 2, g, batch_size=5, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2,
                 compression_scheme=scheme, calibration_threshold_epsilon=0)
    merge_parts(str(tmp_path), "ra.rc1", 2)
    thr = orc.threshold(dark, 0)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16)

    def readers(name, inter):
        out = []
        for off in (False, True):
            rd = ReCoDeReader(str(tmp_path / name), is_intermediate=inter)
            rd.open(print_header=False)
            rd._RA_FRAMES = 4                                            # several batches in this short file
            rd._ra_off = off
            out.append(rd)
        return out
    # get_frame in a loop, then jumps / repeats / backwards
    a, b = readers("ra.rc1", False)
    order = list(range(nz)) + [3, 4, 5, 6, 7, 8, 9, 10, 11, 2, 2, 20, 21, 22, 0, 1, 2, 3]
    for z in order:
        fa, fb = a.get_frame(z), b.get_frame(z)
        assert list(fa) == list(fb) == [z]
        if frames[z].any():
            assert fa[z]["metadata"] == fb[z]["metadata"] and _coo_equal(fa[z]["data"], fb[z]["data"]), "frame %d" % z
            assert np.array_equal(np.asarray(fa[z]["data"].todense()), want[z])
        else:
            assert fa[z]["data"] is None or fa[z]["data"].nnz == 0
            assert (fa[z]["data"] is None) == (fb[z]["data"] is None)
    assert a.readahead_frames_served >= nz - 3 and b.readahead_frames_served == 0
    # get_next_frame from the start to the end of the file, with a get_frame in between
    a2, b2 = readers("ra.rc1", False)
    for z in range(nz):
        if z == 12:
            assert _coo_equal(a2.get_frame(5)[5]["data"], b2.get_frame(5)[5]["data"])
            a2._current_frame_index = b2._current_frame_index = 12
            a2._fp.seek(a2._frame_data_start_position + int(a2._seek_table[12, 1]), 0)
            b2._fp.seek(b2._frame_data_start_position + int(b2._seek_table[12, 1]), 0)
        fa, fb = a2.get_next_frame(), b2.get_next_frame()
        if fa is None or fb is None:                                     # (the empty frame of a reduce-only file; not in these schemes)
            assert fa is None and fb is None
            break
        assert list(fa) == list(fb) == [z]
        if frames[z].any():
            assert _coo_equal(fa[z]["data"], fb[z]["data"])
        assert a2.get_file_position() == b2.get_file_position(), "after frame %d" % z
    assert a2.get_next_frame() is None and b2.get_next_frame() is None      # (end of file: None, as in the reference :224-226)
    assert a2.readahead_frames_served >= nz - 8
    # the part files, sequentially (the reference's own read test): ids as keys, None at the end
    for node in range(2):
        pa, pb = readers("ra.rc1_part%03d" % node, True)
        seen = 0
        while True:
            fa, fb = pa.get_next_frame(), pb.get_next_frame()
            if fb is None:
                assert fa is None
                break
            assert list(fa) == list(fb)
            (fid, body), = fa.items()
            if frames[fid].any():
                assert _coo_equal(body["data"], fb[fid]["data"]) and np.array_equal(np.asarray(body["data"].todense()), want[fid])
            assert pa.get_file_position() == pb.get_file_position()
            seen += 1
        assert seen == len(pa.part_frame_ids) and pa.readahead_frames_served >= seen - 4 and pb.readahead_frames_served == 0
        for r in (pa, pb):
            r.close()
    for r in (a, b, a2, b2):
        r.close()


@pytest.mark.parametrize("scheme", [0, 1])
def test_two_streaming_iterators_side_by_side(scheme, tmp_path, orc):
    """The library's two streaming slots belong to the process, not to a reader: two iterators advanced in turn (two readers of two
    files, as a program comparing datasets would) both deliver every frame - the one that finds a slot taken sends that batch through
    the synchronous call."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, d, nz = 64, 256, 12, 13
    g = load_npz("g3_l1z12.npz")
    sets = []
    for tag, seed in (("one", 5), ("two", 6)):
        dark, frames = synth_frames(seed, nz, ny, nx, 0.04, d)
        sub = tmp_path / tag
        sub.mkdir()
        _write_parts(sub, tag, dark, frames, 1, g, batch_size=4, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=1,
                     compression_scheme=scheme, calibration_threshold_epsilon=0)
        merge_parts(str(sub), tag + ".rc1", 1)
        thr = orc.threshold(dark, 0)
        rd = ReCoDeReader(str(sub / (tag + ".rc1")), is_intermediate=False)
        rd.open(print_header=False)
        sets.append((rd, np.where(frames > thr, frames - thr, 0).astype(np.uint16)))
    its = [rd.iter_frames_triplets(batch=3) for rd, _ in sets]
    seen = [0, 0]
    alive = [True, True]
    while any(alive):
        for j in (0, 1):
            if not alive[j]:
                continue
            try:
                a, pre, tr = next(its[j])
            except StopIteration:
                alive[j] = False
                continue
            for i in range(len(pre) - 1):
                t = tr[int(pre[i]):int(pre[i + 1])]
                dense = np.zeros((ny, nx), np.uint16)
                dense[t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.uint16)
                assert np.array_equal(dense, sets[j][1][a + i]), "reader %d frame %d" % (j, a + i)
                seen[j] += 1
    assert seen == [nz, nz]
    for rd, _ in sets:
        rd.close()


@pytest.mark.parametrize("scheme,level", [(0, 1), (1, 1), (2, 1), (1, 3), (0, 3)])
def test_batched_readers_coo_layout_equals_their_triplets(scheme, level, tmp_path, orc):
    """get_frames_triplets / iter_frames_triplets with coo=True hand out (rows int32, columns int32, values uint16) - the arrays of the COO
    matrices get_frame returns - instead of uint64 triplet rows: same entries, same prefixes, on the device path (zstd, LZ4), the
    host-decoded path (zlib), level 3, a merged file and a part file."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, d, nz = 80, 144, 12, 11
    dark, frames = synth_frames(91, nz, ny, nx, 0.05, d)
    frames[4] = 0
    g = load_npz("g3_l1z12.npz" if level == 1 else "g3_l3z.npz")
    _write_parts(tmp_path, "coo", dark, frames, 2, g, batch_size=3, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2,
                 compression_scheme=scheme, calibration_threshold_epsilon=0, reduction_level=level)
    merge_parts(str(tmp_path), "coo.rc%d" % level, 2)
    for name, inter in (("coo.rc%d" % level, False), ("coo.rc%d_part001" % level, True)):
        rd = ReCoDeReader(str(tmp_path / name), is_intermediate=inter)
        rd.open(print_header=False)
        n = rd._batch_frames()
        p0, t0 = rd.get_frames_triplets(0, n)
        p1, (rows, cols, vals) = rd.get_frames_coo(0, n)
        assert np.array_equal(p0, p1) and rows.dtype == np.int32 and cols.dtype == np.int32 and vals.dtype == np.uint16
        assert np.array_equal(rows, t0[:, 0].astype(np.int32)) and np.array_equal(cols, t0[:, 1].astype(np.int32)) and np.array_equal(vals, t0[:, 2].astype(np.uint16))
        seen = 0
        for a, pre, (r, c, v) in rd.iter_frames_coo(batch=4):
            lo, hi = int(p0[a]), int(p0[a + len(pre) - 1])
            assert np.array_equal(pre - pre[0], p0[a:a + len(pre)] - p0[a])
            assert np.array_equal(r, t0[lo:hi, 0].astype(np.int32)) and np.array_equal(c, t0[lo:hi, 1].astype(np.int32)) and np.array_equal(v, t0[lo:hi, 2].astype(np.uint16))
            seen += len(pre) - 1
        assert seen == n
        rd.close()


@pytest.mark.parametrize("kind", ["level2", "blosc"])
def test_batched_calls_on_files_that_take_the_per_frame_path(kind, tmp_path, orc):
    """Level-2 files and mode-1 files of schemes without a batched decoder (blosc = BASELINE config 4's codec) leave the batched calls
    through their frame-at-a-time fallback: get_frames / get_frames_triplets / get_frames_coo / iter_frames_triplets / iter_frames_coo
    must deliver the same entries as get_frame there, in both layouts (round 3 left the fallback's loop variable named like the
    layout flag)."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, d, nz = 72, 136, 12, 7
    dark, frames = synth_frames(31, nz, ny, nx, 0.05, d)
    frames[nz - 1] = 0                                                    # the LAST frame empty (the case that handed back triplet rows)
    g = load_npz("g3_l1z12.npz")
    level = 2 if kind == "level2" else 1
    over = dict(num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2, compression_scheme=8 if kind == "blosc" else 2,
                reduction_level=level, calibration_threshold_epsilon=0)
    if level == 2:
        over["l2_statistics"] = 2
    _write_parts(tmp_path, "pf", dark, frames, 2, g, batch_size=3, **over)
    merge_parts(str(tmp_path), "pf.rc%d" % level, 2)
    rd = ReCoDeReader(str(tmp_path / ("pf.rc%d" % level)))
    rd.open(print_header=False)
    rd._ra_off = True
    want = []
    for z in range(nz):
        m = rd.get_frame(z)[z]["data"]
        want.append(m if m is not None and m.nnz else None)
    for coo in (False, True):
        prefix, got = rd.get_frames_triplets(0, nz, coo=coo)
        assert rd.last_batch_path == 'per-frame'
        pieces = [(0, prefix, got)]
        pieces += list(rd.iter_frames_triplets(batch=3, coo=coo))
        for a, pre, body in pieces:
            for i in range(len(pre) - 1):
                lo, hi = int(pre[i]), int(pre[i + 1])
                m = want[a + i]
                if m is None:
                    assert hi == lo
                    continue
                if coo:
                    r, c, v = body
                    assert r.dtype == np.int32 and c.dtype == np.int32 and v.dtype == np.uint16
                    assert np.array_equal(r[lo:hi], m.row) and np.array_equal(c[lo:hi], m.col) and np.array_equal(v[lo:hi], m.data.astype(np.uint16))
                else:
                    assert body.dtype == np.uint64 and body.shape[1] == 3
                    assert np.array_equal(body[lo:hi, 0], m.row.astype(np.uint64)) and np.array_equal(body[lo:hi, 1], m.col.astype(np.uint64))
                    assert np.array_equal(body[lo:hi, 2], m.data.astype(np.uint64))
    fr = rd.get_frames(1, nz - 1)
    assert rd.last_batch_path == 'per-frame' and sorted(fr) == list(range(1, nz))
    for z in range(1, nz):
        m = want[z]
        if m is None:
            assert fr[z]["data"].nnz == 0
        else:
            assert np.array_equal(np.asarray(fr[z]["data"].todense()), np.asarray(m.todense()))
    rd.close()


@pytest.mark.parametrize("scheme", [1, 2])
def test_get_frame_out_of_the_readahead_leaves_the_file_where_get_next_frame_needs_it(scheme, tmp_path, orc):
    """get_frame(z) served from the read-ahead window must leave the file position behind frame z, as the frame-at-a-time path does:
    a following get_next_frame() that the read-ahead has nothing for - the next frame is EMPTY, or only one frame is left - reads the
    frame's streams from that position."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, d, nz = 64, 192, 12, 12
    dark, frames = synth_frames(5, nz, ny, nx, 0.04, d)
    frames[7] = 0                                                         # get_frame(6) then get_next_frame() lands on an empty frame
    g = load_npz("g3_l1z12.npz")
    _write_parts(tmp_path, "fp", dark, frames, 1, g, batch_size=4, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=1,
                 compression_scheme=scheme, calibration_threshold_epsilon=0)
    merge_parts(str(tmp_path), "fp.rc1", 1)
    thr = orc.threshold(dark, 0)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16)
    for stop in (6, nz - 2):                                              # next frame empty / next frame the file's last
        rd = ReCoDeReader(str(tmp_path / "fp.rc1"))
        rd.open(print_header=False)
        rd._RA_FRAMES = 4
        ref = ReCoDeReader(str(tmp_path / "fp.rc1"))
        ref.open(print_header=False)
        ref._ra_off = True
        for z in range(stop + 1):
            fa, fb = rd.get_frame(z), ref.get_frame(z)
            if frames[z].any():
                assert _coo_equal(fa[z]["data"], fb[z]["data"])
            assert rd.get_file_position() == ref.get_file_position(), "after get_frame(%d)" % z
        assert rd.readahead_frames_served >= stop - 3
        for z in range(stop + 1, nz):
            fa, fb = rd.get_next_frame(), ref.get_next_frame()
            assert (fa is None) == (fb is None)
            if fa is None:
                break
            assert list(fa) == list(fb) == [z]
            if frames[z].any():
                assert np.array_equal(np.asarray(fa[z]["data"].todense()), want[z]), "frame %d after get_frame(%d)" % (z, stop)
            assert rd.get_file_position() == ref.get_file_position()
        rd.close()
        ref.close()


@pytest.mark.parametrize("scheme", [0, 1])
def test_own_iterator_and_readahead_on_one_reader_do_not_share_batches(scheme, tmp_path, orc):
    """The read-ahead under get_frame keeps an iterator alive on the reader's page-locked batch buffers, with one batch queued on the
    device.  A caller's own iter_frames_* on the SAME reader ends it first and keeps it off while it runs; get_frame calls in between
    return the right frames (frame by frame), and the read-ahead comes back afterwards."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    ny, nx, d, nz = 64, 160, 12, 26
    dark, frames = synth_frames(17, nz, ny, nx, 0.05, d)
    g = load_npz("g3_l1z12.npz")
    _write_parts(tmp_path, "mix", dark, frames, 1, g, batch_size=5, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=1,
                 compression_scheme=scheme, calibration_threshold_epsilon=0)
    merge_parts(str(tmp_path), "mix.rc1", 1)
    thr = orc.threshold(dark, 0)
    want = np.where(frames > thr, frames - thr, 0).astype(np.uint16)
    rd = ReCoDeReader(str(tmp_path / "mix.rc1"))
    rd.open(print_header=False)
    rd._RA_FRAMES = 4

    def dense(z):
        return np.asarray(rd.get_frame(z)[z]["data"].todense())
    for z in range(6):                                                    # the read-ahead is up: a window in hand, the next batch queued
        assert np.array_equal(dense(z), want[z])
    assert rd.readahead_frames_served >= 3
    served = rd.readahead_frames_served
    seen = 0
    for a, pre, (r, c, v) in rd.iter_frames_coo(batch=3):
        r, c, v, pre = r.copy(), c.copy(), v.copy(), pre.copy()
        # frame-at-a-time calls in between: inside the old window, behind it, and sequentially
        for z in (4, 5, 6, 7, 8, 9):
            assert np.array_equal(dense(z), want[z]), "get_frame(%d) next to the iterator's batch at %d" % (z, a)
        for i in range(len(pre) - 1):
            lo, hi = int(pre[i]), int(pre[i + 1])
            img = np.zeros((ny, nx), np.uint16)
            img[r[lo:hi], c[lo:hi]] = v[lo:hi]
            assert np.array_equal(img, want[a + i]), "iterator frame %d" % (a + i)
            seen += 1
    assert seen == nz and rd.readahead_frames_served == served            # nothing came out of a read-ahead while the iterator lived
    for z in range(10, 20):
        assert np.array_equal(dense(z), want[z])
    assert rd.readahead_frames_served > served                            # and it is back afterwards
    rd.close()


@pytest.mark.parametrize("tag,gap", [("u8d8", -1), ("u8d6", -1), ("u8d8v", 2), ("u8cast", -1)])
def test_uint8_sources_reproduce_the_references_files(tag, gap, tmp_path):
    """G10: files the reference wrote from 8-bit sources (uint8 frames and dark; d = 8 raw bytes, d = 6 bit-packed; validation frames;
    frames handed over as uint16 and cast) - part files, validation side files, dose rates and merged files byte for byte, and the
    reader hands back uint8 matrices equal to the reference reader's."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g10_%s.npz" % tag)
    base, nodes = "g10_" + tag, int(g["n_nodes"])
    frames = g["frames"].astype(np.dtype(str(g["given"])))
    for node in range(nodes):
        ip, cfg = _params(tmp_path, g)
        with __import__("warnings").catch_warnings():
            __import__("warnings").simplefilter("ignore")
            w = ReCoDeWriter(base, dark_data=g["dark"], output_directory=str(tmp_path), input_params=ip, mode="batch",
                             validation_frame_gap=gap, node_id=node, batch_size=2)
            w.start()
            m = w.run(frames)
            w.close()
        fn = "%s.rc1_part%03d" % (base, node)
        assert (tmp_path / fn).read_bytes() == open(os.path.join(FILES, fn), "rb").read(), fn
        if gap > 0:
            side = tmp_path / ("%s_part%03d_validation_frames.bin" % (base, node))
            assert np.array_equal(np.fromfile(side, np.uint8), g["vframes%d" % node])
            assert np.allclose(np.asarray(m["run_dose_rates"], np.float64), g["rates%d" % node])
    merged = base + ".rc1"
    merge_parts(str(tmp_path), merged, nodes)
    assert (tmp_path / merged).read_bytes() == open(os.path.join(FILES, merged), "rb").read()
    rd = ReCoDeReader(str(tmp_path / merged))
    rd.open(print_header=False)
    for z in range(g["frames"].shape[0]):
        m = rd.get_frame(z)[z]["data"]
        assert m.dtype == np.uint8 and np.array_equal(np.asarray(m.todense()), g["decoded"][z])
    pre, (rows, cols, vals) = rd.get_frames_coo(0, g["frames"].shape[0])
    img = np.zeros(g["frames"].shape, np.uint8)
    for z in range(g["frames"].shape[0]):
        lo, hi = int(pre[z]), int(pre[z + 1])
        img[z, rows[lo:hi], cols[lo:hi]] = vals[lo:hi]
    assert np.array_equal(img, g["decoded"])
    rd.close()


def test_uint8_sources_with_a_device_codec_round_trip(tmp_path, orc):
    """uint8 frames through the streaming writer with the device's own codecs (zstd modelled, LZ4) and back through the batched reader."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    g = load_npz("g10_u8d8.npz")
    rng = np.random.default_rng(3)
    ny, nx, nz = 96, 200, 9
    dark = rng.integers(1, 9, (ny, nx)).astype(np.uint8)
    frames = np.where(rng.random((nz, ny, nx)) < 0.06, dark + rng.integers(1, 200, (nz, ny, nx)), dark // 2).astype(np.uint8)
    want = np.where(frames > dark, frames - dark, 0).astype(np.uint8)
    for scheme in (1, 2):
        sub = tmp_path / ("s%d" % scheme)
        sub.mkdir()
        _write_parts(sub, "u8", dark, frames, 2, g, batch_size=4, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2,
                     compression_scheme=scheme, calibration_threshold_epsilon=0)
        merge_parts(str(sub), "u8.rc1", 2)
        rd = ReCoDeReader(str(sub / "u8.rc1"))
        rd.open(print_header=False)
        got = np.zeros_like(want)
        for a, pre, (r, c, v) in rd.iter_frames_coo(batch=4):
            for i in range(len(pre) - 1):
                lo, hi = int(pre[i]), int(pre[i + 1])
                got[a + i, r[lo:hi], c[lo:hi]] = v[lo:hi]
        assert rd.last_batch_path == 'device' and np.array_equal(got, want)
        for z in range(nz):
            assert np.array_equal(np.asarray(rd.get_frame(z)[z]["data"].todense()), want[z])
        rd.close()


@pytest.mark.parametrize("given", [np.uint32, np.int64, np.float32])
def test_frames_handed_over_in_another_dtype_are_cast_like_the_reference(given, tmp_path):
    """The reference casts whatever array it is handed to the source dtype its parameters name (recode_writer.py:352-354: `data.astype`);
    so does the staging copy here - G3's 12-bit stack handed over as uint32 / int64 / float32 gives the reference's files byte for byte."""
    import warnings
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g3_l1z12.npz")
    base, nodes = "g3_l1z12", int(g["n_nodes"])
    for node in range(nodes):
        ip, cfg = _params(tmp_path, g)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w = ReCoDeWriter(base, dark_data=g["dark"].astype(given), output_directory=str(tmp_path), input_params=ip, mode="batch",
                             validation_frame_gap=-1, node_id=node, batch_size=3)
            w.start()
            w.run(g["frames"].astype(given))
            w.close()
        fn = "%s.rc1_part%03d" % (base, node)
        assert (tmp_path / fn).read_bytes() == open(os.path.join(FILES, fn), "rb").read(), fn


def test_sources_the_device_does_not_take_are_refused_by_name(tmp_path):
    """Signed sources (source_data_type 1) and reduction level 2 on uint32 sources: refused when the writer is made, with a message that says
    which and why (not a fallback, not a wrong file)."""
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g3_l1z16.npz")
    ip, cfg = _params(tmp_path, g, source_data_type=1, target_data_type=1)
    with pytest.raises(NotImplementedError, match="int16"):
        ReCoDeWriter("x", dark_data=g["dark"], output_directory=str(tmp_path), input_params=ip, mode="batch", node_id=0)
    ip, cfg = _params(tmp_path, g, source_bit_depth=24, target_bit_depth=24, reduction_level=2)
    with pytest.raises(NotImplementedError, match="level 2"):
        ReCoDeWriter("x", dark_data=g["dark"].astype(np.uint32), output_directory=str(tmp_path), input_params=ip, mode="batch", node_id=0)


@pytest.mark.parametrize("tag", ["u32d20", "u32d32", "u32d24", "u32d17", "u32d20v"])
def test_uint32_sources_reproduce_the_references_files(tag, tmp_path):
    """G11: files the reference wrote from sources beyond 16 bits (uint32 frames and dark: 20- and 17-bit fields, four raw bytes a value at
    d = 32 and d = 24) - part files and merged files byte for byte; the reader returns what the reference's reader returned (at d = 24 that
    is NOT what was written: 24-bit fields read out of 32-bit values, the reference's own behaviour - the fixture holds it)."""
    import warnings
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    from pyrecode_amd.recode_writer import ReCoDeWriter
    g = load_npz("g11_%s.npz" % tag)
    base, nodes = "g11_" + tag, int(g["n_nodes"])
    for node in range(nodes):
        ip, cfg = _params(tmp_path, g)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w = ReCoDeWriter(base, dark_data=g["dark"], output_directory=str(tmp_path), input_params=ip, mode="batch",
                             validation_frame_gap=int(g["gap"]), node_id=node, batch_size=2)
            w.start()
            m = w.run(g["frames"])
            w.close()
        fn = "%s.rc1_part%03d" % (base, node)
        assert (tmp_path / fn).read_bytes() == open(os.path.join(FILES, fn), "rb").read(), fn
        if int(g["gap"]) > 0:    # validation frames: uint32 frames in the side file, the device's ROI component counts as dose rates
            side = tmp_path / ("%s_part%03d_validation_frames.bin" % (base, node))
            assert np.array_equal(np.fromfile(side, np.uint8), g["vframes%d" % node])
            assert np.allclose(np.asarray(m["run_dose_rates"], np.float64), g["rates%d" % node])
    merged = base + ".rc1"
    merge_parts(str(tmp_path), merged, nodes)
    assert (tmp_path / merged).read_bytes() == open(os.path.join(FILES, merged), "rb").read()
    rd = ReCoDeReader(str(tmp_path / merged))
    rd.open(print_header=False)
    nz = g["frames"].shape[0]
    for z in range(nz):
        m = rd.get_frame(z)[z]["data"]
        assert m.dtype == np.uint32 and np.array_equal(np.asarray(m.todense()).astype(np.uint64), g["decoded"][z]), "frame %d" % z
    pre, trip = rd.get_frames_triplets(0, nz)                                    # the batched reader: triplet rows hold values of any width
    img = np.zeros(g["frames"].shape, np.uint64)
    for z in range(nz):
        t = trip[int(pre[z]):int(pre[z + 1])]
        img[z, t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2]
    assert np.array_equal(img, g["decoded"])
    rd.close()


@pytest.mark.parametrize("scheme", [1, 2])
def test_uint32_sources_with_a_device_codec_round_trip(scheme, tmp_path):
    """uint32 frames (20 bits) through the streaming writer with zstd / LZ4 on the device and back through both readers."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    g = load_npz("g11_u32d20.npz")
    rng = np.random.default_rng(5)
    ny, nx, nz = 96, 200, 7
    dark = rng.integers(1000, 70000, (ny, nx)).astype(np.uint32)
    frames = np.where(rng.random((nz, ny, nx)) < 0.05, dark + rng.integers(1, 900000, (nz, ny, nx)), dark // 2).astype(np.uint32)
    want = np.where(frames > dark, frames - dark, 0).astype(np.uint32)
    _write_parts(tmp_path, "u32", dark, frames, 2, g, batch_size=3, num_rows=ny, num_cols=nx, num_frames=nz, num_threads=2,
                 compression_scheme=scheme, calibration_threshold_epsilon=0)
    merge_parts(str(tmp_path), "u32.rc1", 2)
    rd = ReCoDeReader(str(tmp_path / "u32.rc1"))
    rd.open(print_header=False)
    for z in range(nz):
        m = rd.get_frame(z)[z]["data"]
        assert m.dtype == np.uint32 and np.array_equal(np.asarray(m.todense()), want[z])
    got = np.zeros_like(want)
    for a, pre, trip in rd.iter_frames_triplets(batch=3):
        for i in range(len(pre) - 1):
            t = trip[int(pre[i]):int(pre[i + 1])]
            got[a + i, t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.uint32)
    assert np.array_equal(got, want)
    rd.close()


def test_random_configurations_round_trip_through_the_public_api(tmp_path):
    """Twenty-four random configurations - source dtype (uint8 / uint16 / uint32) with any source_bit_depth it allows (1 bit on), any frame
    size, one to three nodes, levels 1 and 3, schemes zlib / zstd / LZ4 / blosc-lz4 / none, epsilon 0 - 5, batch sizes that do not divide the
    frame count - through ReCoDeWriter -> part files -> merge_parts -> ReCoDeReader (frame by frame and batched): every decoded frame equals
    where(frame > thr, residual modulo 2^d, 0) (level 3: the binary map)."""
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    g = load_npz("g3_l1z12.npz")
    rng = np.random.default_rng(int(os.environ.get("RC_FUZZ_SEED", "424242")))   # (by hand: other seeds, RC_FUZZ_CASES=200)
    for case in range(int(os.environ.get("RC_FUZZ_CASES", "24"))):
        sb = int(rng.choice([1, 2, 2, 2, 4]))
        d = int(rng.integers(1, 9)) if sb == 1 else (int(rng.integers(9, 17)) if sb == 2 else int(rng.integers(17, 33)))
        dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[sb]
        ny, nx, nz = int(rng.integers(5, 160)), int(rng.integers(8, 200)), int(rng.integers(1, 9))
        if os.environ.get("RC_FUZZ_BIG"):   # (by hand: frames of a hundred tiles and more)
            ny, nx, nz = int(rng.integers(300, 900)), int(rng.integers(400, 1200)), int(rng.integers(1, 6))
        level = int(rng.choice([1, 1, 1, 3] + ([2] if sb != 4 else [])))    # (level 2: uint8 / uint16 sources)
        l2stat = int(rng.integers(0, 3))
        scheme = int(rng.choice([0, 1, 2, 8]))
        mode = int(rng.choice([1, 1, 1, 0]))
        nodes, eps = int(rng.integers(1, 4)), int(rng.integers(0, 6))
        s = float(rng.choice([0.0, 0.004, 0.02, 0.08, 0.3]))
        top = (1 << min(d, 31)) - 1
        dark = rng.integers(0, max(2, min(top // 3, 200)), (ny, nx)).astype(dt)
        amp = rng.integers(1, max(2, min(top, 1 << 20)), (nz, ny, nx))
        frames = np.where(rng.random((nz, ny, nx)) < s, np.minimum(dark.astype(np.int64) + eps + amp, np.iinfo(dt).max), dark // 2).astype(dt)
        if mode == 0 or level == 2:   # reduce-only files (and level 2): the reference's reader takes an EMPTY frame for the end of the file (recode_reader.py:203-213,392-396:
            for z in range(nz):   # get_frame returns None and cuts nz) - mirrored here, so every frame gets an event
                frames[z].flat[z] = min(int(dark.flat[z]) + eps + 1, np.iinfo(dt).max)
        base = "fz%03d" % case
        tag = "case %d: %s d=%d %dx%dx%d level %d scheme %d mode %d nodes %d eps %d s %g" % (case, dt.__name__, d, nz, ny, nx, level, scheme, mode, nodes, eps, s)
        sub = tmp_path / base
        sub.mkdir()
        over = dict(num_rows=ny, num_cols=nx, num_frames=nz, num_threads=nodes, compression_scheme=scheme, calibration_threshold_epsilon=eps,
                    reduction_level=level, rc_operation_mode=mode, source_bit_depth=d, target_bit_depth=d, source_data_type=0, target_data_type=0,
                    l2_statistics=l2stat)
        stream = bool(rng.integers(0, 3) == 0)       # a third of the cases: mode='stream', the frames handed over in random chunks
        gap = int(rng.choice([-1, -1, 2, 3]))        # ... and half of them with validation frames (side file + dose rates)
        cuts = sorted(set(rng.integers(1, nz + 1, int(rng.integers(1, 4))).tolist() + [nz])) if stream else [nz]
        if stream:
            over["num_frames"] = 1                   # (as in fixture G7: no more than the smallest chunk holds - recode_writer.py:284-285; stream mode takes every frame of a chunk)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from pyrecode_amd.recode_writer import ReCoDeWriter
            for node in range(nodes):
                ip, cfg = _params(sub, g, **over)
                w = ReCoDeWriter(base, dark_data=dark, output_directory=str(sub), input_params=ip, mode="stream" if stream else "batch",
                                 validation_frame_gap=gap, node_id=node, run_name=base, batch_size=int(rng.integers(1, 5)))
                w.start()
                at = 0
                for c in cuts:
                    w.run(frames[at:c])
                    at = c
                w.close()
        tag += " %s cuts %s gap %d" % ("stream" if stream else "batch", cuts, gap)
        merged = "%s.rc%d" % (base, level)
        merge_parts(str(sub), merged, nodes)
        thr = (dark.astype(np.int64) + eps) & np.iinfo(dt).max
        if d in (24,) and sb == 4:
            continue    # (24-bit fields of 32-bit values: the reference's reader returns something else - fixture G11 holds that)
        mask = (1 << d) - 1 if d < 32 else 0xFFFFFFFF
        want = np.where(frames.astype(np.int64) > thr, (frames.astype(np.int64) - thr) & mask, 0)
        if level in (2, 3):
            want = (frames.astype(np.int64) > thr).astype(np.int64)
        rd = ReCoDeReader(str(sub / merged))
        rd.open(print_header=False)
        assert rd._header["nz"] == nz, tag + ": the merged file holds %d frames" % rd._header["nz"]
        for z in range(nz):
            try:
                f = rd.get_frame(z)
            except Exception as e:
                raise AssertionError(tag + " get_frame(%d): %r (header nz %d)" % (z, e, rd._header["nz"]))
            m = f[z]["data"] if f is not None else None
            got = np.zeros((ny, nx), np.int64) if m is None else np.asarray(m.todense()).astype(np.int64)
            assert np.array_equal(got, want[z]), tag + " frame %d" % z
            if level == 2 and f is not None:   # one statistic per 8-connected component of the map, scipy label order, modulo 2^d
                import scipy.ndimage as nd
                labels, n = nd.label(want[z] != 0, structure=np.ones((3, 3), int))
                raw = frames[z].astype(np.int64)
                st = nd.sum(raw, labels, np.arange(1, n + 1)) if l2stat == 2 else nd.maximum(raw, labels, np.arange(1, n + 1))
                gs, ws = np.asarray(f[z]["summary_stats"]).astype(np.int64), np.asarray(st, np.int64) & mask
                assert np.array_equal(gs, ws), tag + " frame %d statistics (stat %d): got %d values %s, want %d %s (raw %s)" % (
                    z, l2stat, gs.size, gs[:12].tolist(), ws.size, ws[:12].tolist(), np.asarray(st, np.int64)[:12].tolist())
        rd.close()
        # the part files frame by frame (get_next_frame on intermediate files): every node's frames under their absolute ids
        seen = 0
        for node in range(nodes):
            pr = ReCoDeReader(str(sub / ("%s_part%03d" % (merged, node))), is_intermediate=True)
            pr.open(print_header=False)
            while True:
                f = pr.get_next_frame()
                if f is None:
                    break
                (fid, fd), = f.items()
                assert np.array_equal(np.asarray(fd["data"].todense()).astype(np.int64), want[fid]), tag + " part %d frame id %d" % (node, fid)
                seen += 1
            pr.close()
        assert seen == nz or (s == 0.0 and seen <= nz), tag + ": %d frames in the part files" % seen   # (an empty frame ends a reduce-only part file, as in the reference)
        if level == 2:
            continue
        rd = ReCoDeReader(str(sub / merged))
        rd.open(print_header=False)
        got = np.zeros_like(want)
        for a, pre, trip in rd.iter_frames_triplets(batch=3):
            for i in range(len(pre) - 1):
                t = trip[int(pre[i]):int(pre[i + 1])]
                got[a + i, t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)] = t[:, 2].astype(np.int64)
        assert np.array_equal(got, want), tag + " (batched)"
        # ... the same through get_frames (many frames per device call, COO matrices) from a random start
        z0 = int(rng.integers(0, nz))
        many = rd.get_frames(z0, nz - z0)
        assert sorted(many) == list(range(z0, nz)), tag + " (get_frames keys)"
        for z, fd in many.items():
            assert np.array_equal(np.asarray(fd["data"].todense()).astype(np.int64), want[z]), tag + " (get_frames) frame %d" % z
        # and as the COO layout's three arrays (values beyond 16 bits: the same arrays with uint32 values, through the triplets)
        got = np.zeros_like(want)
        for a, pre, (rows, cols, vals) in rd.iter_frames_coo(batch=int(rng.integers(1, 5))):
            assert rows.dtype == np.int32 and cols.dtype == np.int32 and vals.dtype == (np.uint32 if d > 16 and level == 1 else np.uint16), tag
            for i in range(len(pre) - 1):
                lo, hi = int(pre[i]), int(pre[i + 1])
                got[a + i, rows[lo:hi], cols[lo:hi]] = vals[lo:hi]
        assert np.array_equal(got, want), tag + " (COO layout)"
        pre, (rows, cols, vals) = rd.get_frames_coo(z0, nz - z0)
        got = np.zeros_like(want[z0:])
        for i in range(nz - z0):
            lo, hi = int(pre[i]), int(pre[i + 1])
            got[i, rows[lo:hi], cols[lo:hi]] = vals[lo:hi]
        assert np.array_equal(got, want[z0:]), tag + " (get_frames_coo)"
        rd.close()
