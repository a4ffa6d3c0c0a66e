"""-m gpu: BASELINE.json's full frame sizes.  The oracle's C restatement handles a 4096x4096 frame in ~10 ms, so the full
frames are compared bit-for-bit (not just through properties); on top come the size-independent properties: write -> read
round trip equals where(frame > thr, frame - thr, 0), popcount(bitmap) == nnz == bytes_in_packed_pixvals * 8 // d, metadata
equals stream lengths, and the device generator equals its host mirror."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from pyrecode_amd import _lib as hip, synth
    from oracle import oracle as orc
    orc.lib()
    if hip.device_count() == 0:
        pytest.fail("no GPU visible")
    return torch, hip, synth, orc


def _device_stack(torch, hip, seed, n_frames, N, ppm):
    dark = torch.empty(N, dtype=torch.int16, device="cuda")
    frames = torch.empty((n_frames, N), dtype=torch.int16, device="cuda")
    hip.check(hip.lib().rc_synth_dark(0, seed, N, dark.data_ptr()))
    hip.check(hip.lib().rc_synth_frames(0, seed, 0, n_frames, N, ppm, dark.data_ptr(), frames.data_ptr()))
    return dark, frames


def _stock_lz4f_decompress(data, cap):
    """STOCK liblz4 (LZ4F_decompress), the library behind the reference's lz4.frame (recode_compressors.py:46-49).  Not optional at
    full size: the GPU box has it (bench.py's cpu_baseline links the same library) - a missing library FAILS instead of leaving the
    device's frames to this repo's own from-spec decoder alone."""
    import ctypes as C
    import ctypes.util
    name = ctypes.util.find_library("lz4")
    assert name, "liblz4 not found: the full-size LZ4 records need the stock decoder as their judge"
    L = C.CDLL(name)
    L.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    L.LZ4F_createDecompressionContext.restype = C.c_size_t
    L.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
    L.LZ4F_decompress.restype = C.c_size_t
    L.LZ4F_isError.argtypes = [C.c_size_t]
    L.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
    ctx = C.c_void_p()
    assert not L.LZ4F_isError(L.LZ4F_createDecompressionContext(C.byref(ctx), 100))
    src = np.frombuffer(data, np.uint8)
    dst = np.empty(cap + 64, np.uint8)
    sp = dp = 0
    try:
        while sp < src.size:
            ssz, dsz = C.c_size_t(src.size - sp), C.c_size_t(dst.size - dp)
            r = L.LZ4F_decompress(ctx, dst.ctypes.data + dp, C.byref(dsz), src.ctypes.data + sp, C.byref(ssz), None)
            assert not L.LZ4F_isError(r), "stock liblz4 rejected the frame"
            sp += ssz.value
            dp += dsz.value
            if r == 0:
                break
            assert ssz.value or dsz.value, "stock liblz4 made no progress"
    finally:
        L.LZ4F_freeDecompressionContext(ctx)
    assert sp == src.size, "stock liblz4 did not consume the whole stream"
    return dst[:dp].tobytes()


def _decode(orc, scheme, stream, cap):
    if scheme == "zd":      # compression_scheme 0 from the device's DEFLATE encoder: stdlib zlib, the reference reader's own call
        import zlib
        return zlib.decompress(stream)
    if scheme == 2:
        got = orc.lz4f_decode(stream, cap)
        assert _stock_lz4f_decompress(stream, cap) == got      # both judges, and they agree
        return got
    from pyrecode_amd.recode_compressors import _zstd_host_decompress
    return _zstd_host_decompress(stream)


@pytest.mark.parametrize("depth,ppm,scheme", [(16, 10000, 2), (12, 10000, 2), (16, 1000, 2), (16, 10000, 1), (12, 1000, 1), (16, 10000, "zd"), (12, 10000, "zd")])
def test_4096_device_codec_records_full_oracle_compare(env, depth, ppm, scheme):
    """configs[1] / configs[2] (and d = 12 / 0.1 % variants): 4096x4096 uint16, L1 + LZ4 or zstd, device-resident in and out."""
    _full_oracle_compare(env, 4096, 4096, 6, depth, ppm, scheme)


@pytest.mark.parametrize("depth,ppm,scheme", [(12, 10000, 2), (14, 20000, 1), (12, 10000, "zd")])
def test_3838x3710_frames_full_oracle_compare(env, depth, ppm, scheme):
    """A common detector format whose frames do not end on a bitmap byte (N % 8 = 4; every other frame of a stack starts 8 bytes off a
    16-byte boundary): the tiles inside the frame take the vector-load kernel, the partial last tile the guarded loads."""
    _full_oracle_compare(env, 3710, 3838, 5, depth, ppm, scheme)


@pytest.mark.parametrize("depth,ppm,scheme", [(16, 100000, 2), (12, 300000, 2), (16, 300000, 1), (12, 600000, 2), (16, 600000, 1), (12, 100000, "zd"), (16, 300000, "zd")])
def test_4096_dense_frames_full_oracle_compare(env, depth, ppm, scheme):
    """Beyond the sparse regime (10 %, 30 %, 60 % of the pixels set): tiles past the staged compaction's capacity in every frame, blocks
    stored raw, residual slots instead of combined slots, records of 4 - 22 MB - still the reference's bytes (recode_writer.py:437-440,
    518-525), judged like the sparse ones."""
    _full_oracle_compare(env, 4096, 4096, 3, depth, ppm, scheme)


def test_4096_all_set_frame_is_refused_like_the_reference(env):
    """A frame whose every pixel is above threshold: bitmap stream + 2 N bytes of residuals exceed the raw frame, and the reference raises
    ValueError('Buffer size smaller than compressed data size') (recode_writer.py:565-566).  The device flags the batch
    (RC_ERR_RECORD_TOO_LARGE, naming the frame); the frames in front of it in the same batch are unaffected when run alone."""
    torch, hip, synth, orc = env
    ny = nx = 4096
    N, B = ny * nx, 3
    dark_d, frames_d = _device_stack(torch, hip, 5, B, N, 10000)
    # every pixel far above the dark level (80..120), with residuals no entropy coder can shrink: a CONSTANT all-set frame is a legal
    # record under zstd (its residual stream Huffman-codes to half - the device wrote 19 MB for it), under LZ4 it is not
    frames_d[1] = torch.randint(2000, 62000, (N,), device="cuda", dtype=torch.int32).to(torch.int16)   # (wraps to the same 16 bits)
    for scheme in (2, 1):
        ctx = hip.ReduceContext(nx, ny, 16, 1, 1, scheme, 1, 0, max_batch=B)
        ctx.set_dark(dark_d.data_ptr(), 0)
        cap = int(hip.lib().rc_out_capacity(ctx.handle, B))
        out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        rec = torch.empty(B + 1, dtype=torch.int64, device="cuda")
        md = torch.empty((B, 3), dtype=torch.int32, device="cuda")
        ctx.enqueue(frames_d.data_ptr(), B, 0, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
        with pytest.raises(ValueError, match="Buffer size smaller than compressed data size") as ei:
            ctx.sync()
        assert "frame 1" in str(ei.value)
        # the same context takes the next (legal) batch
        ctx.enqueue(frames_d[2:].data_ptr(), 1, 7, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
        ctx.sync()
        rec_h = rec.cpu().numpy()
        r = out[:int(rec_h[1])].cpu().numpy().tobytes()
        fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        thr = dark_d.cpu().numpy().view(np.uint16)
        bitmap, packed, nnz = orc.reduce_frame_l1(frames_d[2].cpu().numpy().view(np.uint16), thr, 16)
        assert fid == 7 and npk == packed.size and len(r) == 16 + cb + cp
        assert _decode(orc, scheme, r[16:16 + cb], bitmap.size + 8) == bitmap.tobytes()
        assert _decode(orc, scheme, r[16 + cb:], packed.size + 8) == packed.tobytes()
        ctx.close()


def _full_oracle_compare(env, ny, nx, B, depth, ppm, scheme):
    torch, hip, synth, orc = env
    N = ny * nx
    dark_d, frames_d = _device_stack(torch, hip, 7, B, N, ppm)
    ctx = hip.ReduceContext(nx, ny, depth, 1, 1, 0 if scheme == "zd" else scheme, 1, 0, max_batch=B, device_zlib=scheme == "zd")
    ctx.set_dark(dark_d.data_ptr(), 0)
    cap = int(hip.lib().rc_out_capacity(ctx.handle, B))   # B raw frames: what a batch of legal records cannot exceed
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    rec = torch.empty(B + 1, dtype=torch.int64, device="cuda")
    md = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    ctx.enqueue(frames_d.data_ptr(), B, 1000, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
    ctx.sync()
    rec_h, md_h = rec.cpu().numpy(), md.cpu().numpy()
    out_h = out[:int(rec_h[-1])].cpu().numpy()
    thr = dark_d.cpu().numpy().view(np.uint16)
    frames = frames_d.cpu().numpy().view(np.uint16)
    # generator: device == host mirror (one frame is enough at this size; small sizes are covered elsewhere)
    assert np.array_equal(thr, synth.dark_frame(7, N))
    assert np.array_equal(frames[2], synth.frames(7, 2, 1, N, ppm, thr)[0])
    for z in range(B):
        r = out_h[int(rec_h[z]):int(rec_h[z + 1])].tobytes()
        fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        assert fid == 1000 + z and (cb, cp, npk) == tuple(int(v) for v in md_h[z]) and len(r) == 16 + cb + cp
        bitmap, packed, nnz = orc.reduce_frame_l1(frames[z], thr, depth)
        assert npk == packed.size == (nnz * depth + 7) // 8
        assert _decode(orc, scheme, r[16:16 + cb], bitmap.size + 8) == bitmap.tobytes()
        assert _decode(orc, scheme, r[16 + cb:], packed.size + 8) == packed.tobytes()
        assert int(np.unpackbits(bitmap).sum()) == nnz
        assert abs(nnz / N - ppm / 1e6) < max(2e-4, 6 * (ppm / 1e6 * (1 - ppm / 1e6) / N) ** 0.5)
    ctx.close()


def test_direct_electron_size_reduce_only_and_round_trip(env):
    """configs[4] shape: 11520x8184 uint16, 5 % sparsity, d = 12; reduce-only records vs the oracle, then the reader-side
    expand must give back where(frame > thr, frame - thr, 0)."""
    torch, hip, synth, orc = env
    ny, nx, depth = 8184, 11520, 12
    N, B = ny * nx, 2
    dark_d, frames_d = _device_stack(torch, hip, 11, B, N, 50000)
    ctx = hip.ReduceContext(nx, ny, depth, 1, 0, 0, 1, 0, max_batch=B)
    ctx.set_dark(dark_d.data_ptr(), 3)
    thr = (dark_d.cpu().numpy().view(np.uint16) + np.uint16(3)).astype(np.uint16)
    frames = frames_d.cpu().numpy().view(np.uint16)
    out, rec, md = ctx.reduce_compress_batch(frames.reshape(B, ny, nx), first_frame_id=0)
    nb = (N + 7) // 8
    for z in range(B):
        r = out[int(rec[z]):int(rec[z + 1])]
        bitmap, packed, nnz = orc.reduce_frame_l1(frames[z], thr, depth)
        assert struct.unpack_from("<II", r[:8].tobytes(), 0) == (z, packed.size)
        assert np.array_equal(r[8:8 + nb], bitmap)
        assert np.array_equal(r[8 + nb:], packed)
        # expand (reader side) and compare the dense residual image
        trip = np.empty((nnz, 3), np.uint64)
        got = hip.lib().rc_unpack_frame_sparse(nx, ny, depth, hip.ptr(bitmap), hip.ptr(packed), packed.size, hip.ptr(trip), nnz, 1)
        assert got == nnz
        dense = np.zeros(N, np.uint16)
        dense[(trip[:, 0] * nx + trip[:, 1]).astype(np.int64)] = trip[:, 2].astype(np.uint16)
        want = np.where(frames[z] > thr, frames[z] - thr, 0).astype(np.uint16)
        assert np.array_equal(dense, want)
    ctx.close()


def test_4096_host_buffer_path_and_pcie_inclusive_rate(env, capsys):
    """The synchronous entry point with pageable host frames (what ReCoDeWriter.run uses): same records as the device-resident
    path; prints the PCIe-inclusive rate quoted in DESIGN.md (not bench.py's value)."""
    import time
    torch, hip, synth, orc = env
    ny = nx = 4096
    N, B = ny * nx, 16
    dark_d, frames_d = _device_stack(torch, hip, 3, B, N, 10000)
    frames = frames_d.cpu().numpy().view(np.uint16).reshape(B, ny, nx)
    ctx = hip.ReduceContext(nx, ny, 16, 1, 1, 2, 1, 0, max_batch=B)
    ctx.set_dark(dark_d.data_ptr(), 0)
    out, rec, md = ctx.reduce_compress_batch(frames, 0)           # warm-up + staging allocation
    t0 = time.perf_counter()
    out, rec, md = ctx.reduce_compress_batch(frames, 0, out=out)
    dt = time.perf_counter() - t0
    thr = dark_d.cpu().numpy().view(np.uint16)
    for z in (0, B - 1):
        r = out[int(rec[z]):int(rec[z + 1])].tobytes()
        _, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        bitmap, packed, nnz = orc.reduce_frame_l1(frames[z].ravel(), thr, 16)
        assert _decode(orc, 2, r[16:16 + cb], bitmap.size + 8) == bitmap.tobytes()
        assert _decode(orc, 2, r[16 + cb:], packed.size + 8) == packed.tobytes()
    with capsys.disabled():
        print("\n[pcie-inclusive] %d frames 4096x4096 from pageable host memory: %.1f frames/s (%.2f GB/s in)" % (
            B, B / dt, B * N * 2 / dt / 1e9))
    ctx.close()


@pytest.mark.parametrize("depth", [16, 12])   # 16: BASELINE's "uint16" and bench.py --config 4; 12: the packed form
def test_4096_l2_blosc_config4(env, depth):
    """configs[3]: 4096x4096 uint16, 0.1 % sparsity, L2 (component maxima) + blosc-lz4.  No oracle exists in the reference for
    L2: the checker is the stated intent, scipy.ndimage.label (8-connectivity) + per-label maximum of the raw frame."""
    import scipy.ndimage as nd
    torch, hip, synth, orc = env
    ny = nx = 4096
    N, B = ny * nx, 3
    dark_d, frames_d = _device_stack(torch, hip, 21, B, N, 1000)
    ctx = hip.ReduceContext(nx, ny, depth, 2, 1, 8, 1, 0, max_batch=B)
    ctx.set_dark(dark_d.data_ptr(), 0)
    ctx.set_l2_statistics(1)
    cap = B * (N // 4)
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    rec = torch.empty(B + 1, dtype=torch.int64, device="cuda")
    md = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    ctx.enqueue(frames_d.data_ptr(), B, 0, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
    ctx.sync()
    rec_h = rec.cpu().numpy()
    out_h = out[:int(rec_h[-1])].cpu().numpy()
    thr = dark_d.cpu().numpy().view(np.uint16).reshape(ny, nx)
    frames = frames_d.cpu().numpy().view(np.uint16).reshape(B, ny, nx)
    for z in range(B):
        r = out_h[int(rec_h[z]):int(rec_h[z + 1])].tobytes()
        fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        binary = frames[z] > thr
        labels, n = nd.label(binary, structure=np.ones((3, 3), int))
        vals = np.asarray(nd.maximum(frames[z].astype(np.int64), labels, np.arange(1, n + 1)), np.int64).astype(np.uint16)
        assert orc.blosc1_decode(r[16:16 + cb]) == orc.pack_binary_frame(binary).tobytes()
        assert npk == (n * depth + 7) // 8
        assert orc.blosc1_decode(r[16 + cb:]) == orc.bit_pack(vals, depth).tobytes()
    ctx.close()


def test_direct_electron_size_zstd_write_read_config5(env, tmp_path):
    """configs[4]: 11520x8184 uint16, 5 % sparsity, L1 + zstd, through the reference-shaped API with 2 writers (the contiguous
    block rule), direct merge, and a round trip through ReCoDeReader that must be bit-exact."""
    torch, hip, synth, orc = env
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_writer import ReCoDeWriter
    from pyrecode_amd.recode_reader import ReCoDeReader
    from pyrecode_amd import parallel
    ny, nx, depth, nz = 8184, 11520, 12, 3
    N = ny * nx
    dark_d, frames_d = _device_stack(torch, hip, 13, nz, N, 50000)
    dark = dark_d.cpu().numpy().view(np.uint16).reshape(ny, nx)
    frames = frames_d.cpu().numpy().view(np.uint16).reshape(nz, ny, nx)
    del frames_d
    cfg = dict(l4_centroiding=0, source_file_type=0, num_frames=nz, source_header_length=0, calibration_frame_offset=0,
               compression_scheme=1, calibration_file_type=0, compression_level=1, l2_statistics=0,
               calibration_threshold_epsilon=0, frame_offset=0, num_threads=2, rc_operation_mode=1, num_calibration_frames=1,
               reduction_level=1, keep_calibration_data=1, source_bit_depth=depth, target_bit_depth=depth, keep_part_files=0,
               num_rows=ny, num_cols=nx, source_data_type=0, target_data_type=0)
    pf = tmp_path / "p.txt"
    pf.write_text("".join("%s = %d\n" % kv for kv in cfg.items()))
    for node in range(2):
        ip = InputParams()
        ip.load(str(pf))
        w = ReCoDeWriter("de", dark_data=dark, output_directory=str(tmp_path), input_params=ip, node_id=node, batch_size=2)
        w.start()
        w.run(frames)
        w.close()
    recs = []
    for node in range(2):
        recs += parallel.read_part_records(str(tmp_path / ("de.rc1_part%03d" % node)))[1]
    assert parallel.merge_direct(str(tmp_path), "de.rc1", rank=0, world=1, records=recs) == nz
    rd = ReCoDeReader(str(tmp_path / "de.rc1"))
    rd.open(print_header=False)
    for z in (2, 0):
        got = np.asarray(rd.get_frame(z)[z]["data"].todense())
        want = np.where(frames[z] > dark, frames[z] - dark, 0).astype(np.uint16)
        assert np.array_equal(got, want)
    # the batched path (rc_expand_frames: device zstd decode of both streams + expand, all frames in one call)
    prefix, trip = rd.get_frames_triplets(0, nz)
    assert rd.last_batch_path == "device"
    for z in range(nz):
        t = trip[int(prefix[z]):int(prefix[z + 1])]
        want = np.where(frames[z] > dark, frames[z] - dark, 0).astype(np.uint16)
        rows, cols = np.nonzero(want)
        assert np.array_equal(t[:, 0], rows.astype(np.uint64)) and np.array_equal(t[:, 1], cols.astype(np.uint64))
        assert np.array_equal(t[:, 2], want[rows, cols].astype(np.uint64))
    # the same batch in the COO layout (rc_expand_frames_coo; 14 million set pixels: rows | columns | values each that long), and
    # streamed two frames at a time
    p2, (r2, c2, v2) = rd.get_frames_triplets(0, nz, coo=True)
    assert np.array_equal(p2, prefix)
    assert np.array_equal(r2, trip[:, 0].astype(np.int32)) and np.array_equal(c2, trip[:, 1].astype(np.int32)) and np.array_equal(v2, trip[:, 2].astype(np.uint16))
    for a, pre, (r3, c3, v3) in rd.iter_frames_triplets(batch=2, coo=True):
        lo, hi = int(prefix[a]), int(prefix[a + len(pre) - 1])
        assert np.array_equal(r3, r2[lo:hi]) and np.array_equal(c3, c2[lo:hi]) and np.array_equal(v3, v2[lo:hi])
    rd.close()


@pytest.mark.parametrize("scheme", [2, 1])
def test_4096_detector_like_clusters_full_oracle_compare(env, scheme):
    """The one real-data anchor the reference records (examples/Reading_ReCoDe_v0.1_Files.ipynb cells 7 / 17: 4096^2, 12 bit,
    ~4.3 % of the pixels set): events in 1..6-pixel clusters.  Exercises what Bernoulli frames never reach at speed - tiles
    beyond the staged compaction's capacity (the dense-tile path) and literal-heavy bitmap blocks - frame for frame against
    the oracle, then through the batched device reader."""
    torch, hip, synth, orc = env
    ny = nx = 4096
    N, B, depth, seed_ppm = ny * nx, 4, 12, 11000
    L = hip.lib()
    dark_d = torch.empty(N, dtype=torch.int16, device="cuda")
    frames_d = torch.empty((B, N), dtype=torch.int16, device="cuda")
    hip.check(L.rc_synth_dark(0, 23, N, dark_d.data_ptr()))
    hip.check(L.rc_synth_frames_clustered(0, 23, 0, B, nx, ny, seed_ppm, dark_d.data_ptr(), frames_d.data_ptr()))
    thr = dark_d.cpu().numpy().view(np.uint16)
    frames = frames_d.cpu().numpy().view(np.uint16)
    assert np.array_equal(frames[1], synth.frames_clustered(23, 1, 1, nx, ny, seed_ppm, thr)[0])     # device generator == host mirror
    # make two tiles of frame 0 denser than the staged compaction holds (> 1536 set pixels of 4096) and one tile solid
    frames[0, 5 * 4096:6 * 4096:2] = thr[5 * 4096:6 * 4096:2] + 9
    frames[0, 77 * 4096:78 * 4096] = thr[77 * 4096:78 * 4096] + 1
    frames_d.copy_(torch.from_numpy(frames.view(np.int16)))
    ctx = hip.ReduceContext(nx, ny, depth, 1, 1, scheme, 1, 0, max_batch=B)
    ctx.set_dark(dark_d.data_ptr(), 0)
    cap = B * N
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    rec = torch.empty(B + 1, dtype=torch.int64, device="cuda")
    md = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    ctx.enqueue(frames_d.data_ptr(), B, 0, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
    ctx.sync()
    rec_h, md_h = rec.cpu().numpy(), md.cpu().numpy()
    out_h = out[:int(rec_h[-1])].cpu().numpy()
    want_trip, blobs = [], []
    for z in range(B):
        r = out_h[int(rec_h[z]):int(rec_h[z + 1])].tobytes()
        fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        bitmap, packed, nnz = orc.reduce_frame_l1(frames[z], thr, depth)
        assert fid == z and npk == packed.size and len(r) == 16 + cb + cp
        assert 0.035 < nnz / N < 0.05
        assert _decode(orc, scheme, r[16:16 + cb], bitmap.size + 8) == bitmap.tobytes()
        assert _decode(orc, scheme, r[16 + cb:], packed.size + 8) == packed.tobytes()
        want_trip.append(orc.unpack_frame_sparse(nx, ny, depth, bitmap, packed, 1))
        blobs.append(np.frombuffer(r[16:], np.uint8))
    ctx.close()
    # reader: both streams of all frames decoded and expanded in one device call
    blob = np.ascontiguousarray(np.concatenate(blobs))
    sizes = np.ascontiguousarray(md_h.astype(np.uint32))
    prefix = np.zeros(B + 1, np.uint64)
    capt = int(sum(t.shape[0] for t in want_trip))
    trip = np.empty((capt, 3), np.uint64)
    hip.check(L.rc_expand_frames(nx, ny, depth, 1, 1, scheme, hip.ptr(blob), hip.ptr(sizes), B, hip.ptr(prefix), hip.ptr(trip), capt), "rc_expand_frames")
    assert int(prefix[B]) == capt
    for z in range(B):
        assert np.array_equal(trip[int(prefix[z]):int(prefix[z + 1])], want_trip[z]), "frame %d" % z


@pytest.mark.parametrize("scheme,depth", [(1, 12), (1, 16), (2, 16), (0, 12)])   # (1, 16): Huffman-coded residuals (two second-stage branches)
def test_frames_with_more_than_4096_tiles_take_the_segmented_scans(env, scheme, depth):
    """ntiles = 4200 (two scan segments, k_scan_seg / k_scan_fix): an ordinary frame, a frame whose first 4150 tiles are EMPTY
    (the first block that needs the zstd tree and tables lies in the second segment; every next-non-empty link of the first
    segment crosses the border), an empty frame, and a frame with events only in tile 0 and the last tile - all against the oracle."""
    torch, hip, synth, orc = env
    ny, nx = 4200, 4096
    N, B = ny * nx, 4
    L = hip.lib()
    dark_d = torch.empty(N, dtype=torch.int16, device="cuda")
    frames_d = torch.empty((B, N), dtype=torch.int16, device="cuda")
    hip.check(L.rc_synth_dark(0, 41, N, dark_d.data_ptr()))
    hip.check(L.rc_synth_frames(0, 41, 0, B, N, 10000, dark_d.data_ptr(), frames_d.data_ptr()))
    thr = dark_d.cpu().numpy().view(np.uint16)
    frames = frames_d.cpu().numpy().view(np.uint16).copy()
    frames[1, :4150 * 4096] = thr[:4150 * 4096] // 2
    frames[2] = thr // 2
    frames[3] = thr // 2
    frames[3, 5] = thr[5] + 77
    frames[3, N - 3] = thr[N - 3] + 1234
    frames_d.copy_(torch.from_numpy(frames.view(np.int16)))
    mode = 0 if scheme == 0 else 1
    ctx = hip.ReduceContext(nx, ny, depth, 1, mode, scheme, 1, 0, max_batch=B)
    ctx.set_dark(dark_d.data_ptr(), 0)
    cap = B * N
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    rec = torch.empty(B + 1, dtype=torch.int64, device="cuda")
    md = torch.empty((B, 3), dtype=torch.int32, device="cuda")
    for _ in range(2):   # (zstd: the first batch fits the model; the second runs with it in place)
        ctx.enqueue(frames_d.data_ptr(), B, 0, out.data_ptr(), cap, rec.data_ptr(), md.data_ptr())
        ctx.sync()
        rec_h = rec.cpu().numpy()
        out_h = out[:int(rec_h[-1])].cpu().numpy()
        for z in range(B):
            r = out_h[int(rec_h[z]):int(rec_h[z + 1])].tobytes()
            bitmap, packed, nnz = orc.reduce_frame_l1(frames[z], thr, depth)
            if mode == 0:
                assert struct.unpack_from("<II", r, 0) == (z, packed.size) and r[8:] == bitmap.tobytes() + packed.tobytes(), "frame %d" % z
                continue
            fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
            assert fid == z and npk == packed.size and len(r) == 16 + cb + cp
            assert _decode(orc, scheme, r[16:16 + cb], bitmap.size + 8) == bitmap.tobytes(), "frame %d" % z
            assert _decode(orc, scheme, r[16 + cb:], packed.size + 8) == packed.tobytes(), "frame %d" % z
    ctx.close()


def test_direct_electron_size_device_zlib_write_read(env, tmp_path):
    """configs[4]'s shape with compression_scheme 0 from the device's DEFLATE encoder (ReCoDeWriter(device_zlib=True)): every stream of the
    part files inflates under stdlib zlib to the oracle's bytes, and the reference-shaped reader (zlib.decompress per stream,
    recode_compressors.py:43) gives back where(frame > thr, frame - thr, 0) - frame at a time and batched."""
    import zlib
    torch, hip, synth, orc = env
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_writer import ReCoDeWriter
    from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
    from pyrecode_amd import parallel
    ny, nx, depth, nz = 8184, 11520, 12, 3
    N = ny * nx
    dark_d, frames_d = _device_stack(torch, hip, 17, nz, N, 50000)
    dark = dark_d.cpu().numpy().view(np.uint16).reshape(ny, nx)
    frames = frames_d.cpu().numpy().view(np.uint16).reshape(nz, ny, nx)
    del frames_d
    cfg = dict(l4_centroiding=0, source_file_type=0, num_frames=nz, source_header_length=0, calibration_frame_offset=0,
               compression_scheme=0, calibration_file_type=0, compression_level=1, l2_statistics=0,
               calibration_threshold_epsilon=0, frame_offset=0, num_threads=2, rc_operation_mode=1, num_calibration_frames=1,
               reduction_level=1, keep_calibration_data=1, source_bit_depth=depth, target_bit_depth=depth, keep_part_files=1,
               num_rows=ny, num_cols=nx, source_data_type=0, target_data_type=0)
    pf = tmp_path / "p.txt"
    pf.write_text("".join("%s = %d\n" % kv for kv in cfg.items()))
    for node in range(2):
        ip = InputParams()
        ip.load(str(pf))
        w = ReCoDeWriter("dz", dark_data=dark, output_directory=str(tmp_path), input_params=ip, node_id=node, batch_size=2, device_zlib=True)
        w.start()
        assert w._ctx.on_device_codec and not w._host_compress
        w.run(frames)
        w.close()
    thr = dark.ravel()
    seen = 0
    for node in range(2):
        hdr, recs = parallel.read_part_records(str(tmp_path / ("dz.rc1_part%03d" % node)))
        assert hdr.as_dict()["compression_scheme"] == 0
        for fid, md, data in recs:
            cb, cp, npk = (int(v) for v in md[:3])
            bitmap, packed, nnz = orc.reduce_frame_l1(frames[fid].ravel(), thr, depth)
            assert zlib.decompress(bytes(data[:cb])) == bitmap.tobytes() and zlib.decompress(bytes(data[cb:cb + cp])) == packed.tobytes() and npk == packed.size
            seen += 1
    assert seen == nz
    merge_parts(str(tmp_path), "dz.rc1", 2)
    rd = ReCoDeReader(str(tmp_path / "dz.rc1"))
    rd.open(print_header=False)
    for z in (1, 0, 2):
        got = np.asarray(rd.get_frame(z)[z]["data"].todense())
        assert np.array_equal(got, np.where(frames[z] > dark, frames[z] - dark, 0).astype(np.uint16))
    prefix, trip = rd.get_frames_triplets(0, nz)
    for z in range(nz):
        t = trip[int(prefix[z]):int(prefix[z + 1])]
        want = np.where(frames[z] > dark, frames[z] - dark, 0).astype(np.uint16)
        rows, cols = np.nonzero(want)
        assert np.array_equal(t[:, 0], rows.astype(np.uint64)) and np.array_equal(t[:, 1], cols.astype(np.uint64))
        assert np.array_equal(t[:, 2], want[rows, cols].astype(np.uint64))
    rd.close()
