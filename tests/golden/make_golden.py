#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

Test infrastructure. Run once here (where /root/reference exists); the GPU box only ever sees the
committed outputs (`*.npz`, `files/*`).  Nothing from the reference is copied: fixtures hold
seeded inputs and the reference's outputs only.

How the reference is run (SURVEY.md §8c recipe):
  * oracle/build_ref.sh compiles the reference's own CPython extension `c_recode` from
    /root/reference/pyrecode/pyrecode.cpp (+ c_extensions/reader.h) into oracle/_ref/.
  * `numba` is not installed, so an identity-`jit` stand-in (oracle/_ref/shim/numba) lets
    `pyrecode.recode_writer` import; its @jit kernels (_pack_binary_frame, _bit_pack) then run as
    the plain Python they are written in.  Reader-side modules need no stand-in.
  * ReCoDeServer is not used (zmq absent; np.bool/np.int at recode_server.py:410-411): one
    ReCoDeWriter per node_id is driven in-process, then merge_parts.

Fixture groups:
  G1  thr / binary map / L1 residuals          (recode_writer.py:126-137, 437, 440)
  G2  _pack_binary_frame / _bit_pack outputs   (recode_writer.py:622-634, 637-652)
  G3  whole part files + merged file           (recode_writer.py:184-607, recode_reader.py:495-595)
  G7  mode='stream': part files of a writer fed chunk by chunk (recode_writer.py:311-322,422-423)
  G8  validation frames: side file + dose rates      (recode_writer.py:207-211,400-415)
  G9  bz2 / lzma / zlib level 9 / level 3 reduce-only: whole files (recode_compressors.py:82-101)
  G10 8-bit sources (source dtype uint8, misc.py:41-49): d = 8 (raw bytes) and d = 6 (bit-packed), whole files, one with validation frames
  G11 sources beyond 16 bits (source dtype uint32): d = 20 (bit-packed), d = 32 and d = 24 (`.tobytes()`: four bytes a value), whole files
  G4  512-byte header bytes                    (recode_header.py:58-94, 257-275)
  G5  get_frame_sparse triplets                (pyrecode.cpp:95-119, reader.h:10-68)
  G6  321-byte v0.1 header                     (recode_header.py:27-56, 98-127, 257-275)
"""
import contextlib
import io
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("RECODE_REFERENCE", "/root/reference")
REFBUILD = os.path.join(REPO, "oracle", "_ref")

subprocess.check_call([os.path.join(REPO, "oracle", "build_ref.sh")])
sys.path[:0] = [os.path.join(REFBUILD, "shim"), REFBUILD, REF]

with contextlib.redirect_stdout(io.StringIO()):
    import pyrecode.recode_writer as ref_writer
    import pyrecode.recode_reader as ref_reader
    from pyrecode.params import InputParams
    from pyrecode.recode_header import ReCoDeHeader
    import c_recode as ref_c

# where the fixtures go: tests/golden itself, or RC_GOLDEN_OUT (tests/test_oracle_golden.py::test_committed_fixtures_are_what_the_reference_writes_today
# regenerates everything into a scratch directory and compares it with the committed set)
OUT = os.environ.get("RC_GOLDEN_OUT") or HERE
FILES = os.path.join(OUT, "files")
os.makedirs(FILES, exist_ok=True)

PARAM_DEFAULTS = dict(
    l4_centroiding=0, source_file_type=0, num_frames=8, source_header_length=0,
    calibration_frame_offset=0, compression_scheme=0, calibration_file_type=0, compression_level=1,
    l2_statistics=0, calibration_threshold_epsilon=0, frame_offset=0, num_threads=3,
    rc_operation_mode=1, num_calibration_frames=1, reduction_level=1, keep_calibration_data=1,
    source_bit_depth=12, target_bit_depth=12, keep_part_files=0, num_rows=40, num_cols=56,
    source_data_type=0, target_data_type=0)


def make_params(tmp, **over):
    cfg = dict(PARAM_DEFAULTS)
    cfg.update(over)
    path = os.path.join(tmp, "params.txt")
    with open(path, "w") as f:
        for k, v in cfg.items():
            f.write("%s = %d\n" % (k, v))
    ip = InputParams()
    ip.load(path)
    return ip, cfg


def synth_stack(seed, nz, ny, nx, sparsity, depth, dark_lo=3, dark_hi=20):
    """Seeded frames: Bernoulli(sparsity) events above a random dark level, background <= thr."""
    rng = np.random.default_rng(seed)
    dark = rng.integers(dark_lo, dark_hi + 1, (ny, nx)).astype(np.uint16)
    top = (1 << depth) - 1
    frames = np.empty((nz, ny, nx), np.uint16)
    for z in range(nz):
        mask = rng.random((ny, nx)) < sparsity
        amp = rng.integers(1, top - dark_hi - 8, (ny, nx)).astype(np.uint16)
        below = np.floor(rng.random((ny, nx)) * (dark + 1)).astype(np.uint16)
        frames[z] = np.where(mask, dark + amp, below)
    return dark, frames


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# --------------------------------------------------------------------------------------------
# G1 + G2: per-frame reduce pieces straight from the reference's expressions / kernels
# --------------------------------------------------------------------------------------------
def g1_g2():
    out = {}
    cases = [("a", 37, 53, 0.10, 12, 0), ("b", 40, 56, 0.05, 12, 7), ("c", 64, 64, 0.01, 10, 0),
             ("d", 16, 24, 0.50, 14, 3), ("e", 9, 7, 0.30, 9, 0), ("f", 32, 128, 0.02, 16, 0)]
    for tag, ny, nx, s, depth, eps in cases:
        dark, frames = synth_stack(1000 + len(out), 2, ny, nx, s, depth)
        if tag == "f":  # exercise 65535 and exact-threshold pixels at depth 16
            frames[0, 0, 0] = 65535
            frames[0, 0, 1] = dark[0, 1] + eps          # == thr  -> NOT foreground (strict >)
            frames[0, 0, 2] = dark[0, 2] + eps + 1      # thr + 1 -> residual 1
        # recode_writer.py:127 (NumPy 2: uint16 + python int stays uint16)
        thr = dark + eps
        out[f"g1_{tag}_dark"] = dark
        out[f"g1_{tag}_eps"] = np.int64(eps)
        out[f"g1_{tag}_depth"] = np.int64(depth)
        out[f"g1_{tag}_frames"] = frames
        out[f"g1_{tag}_thr"] = thr
        for z in range(frames.shape[0]):
            frame = frames[z]
            binary = frame > thr                              # recode_writer.py:437
            pix = frame[binary] - thr[binary]                 # recode_writer.py:440
            nb = int(np.ceil(ny * nx / 8))
            packed_map = ref_writer._pack_binary_frame(binary, nb)   # recode_writer.py:622-634
            out[f"g1_{tag}_binary{z}"] = binary
            out[f"g1_{tag}_pix{z}"] = pix
            out[f"g2_{tag}_bitmap{z}"] = np.asarray(packed_map, np.uint8)
            if depth % 8 != 0:
                out[f"g2_{tag}_packed{z}"] = np.asarray(ref_writer._bit_pack(pix, depth), np.uint8)
            else:
                out[f"g2_{tag}_packed{z}"] = np.frombuffer(pix.tobytes(), np.uint8)  # :463-464
    # _bit_pack on its own for every depth, including values with bits above the depth (dropped)
    rng = np.random.default_rng(77)
    vals = rng.integers(0, 65536, 41).astype(np.uint16)
    out["g2_vals"] = vals
    for d in (1, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16):
        out[f"g2_bitpack_d{d}"] = np.asarray(ref_writer._bit_pack(vals, d), np.uint8)
    # the reference C packer called directly (first call into a zeroed buffer == intended semantics)
    np.savez_compressed(os.path.join(OUT, "g1_g2_reduce.npz"), **out)
    print("g1_g2:", len(out), "arrays")


# --------------------------------------------------------------------------------------------
# G3 + G4: whole files
# --------------------------------------------------------------------------------------------
def write_parts(tmp, base, dark, frames, n_nodes, **over):
    nz, ny, nx = frames.shape
    ip, cfg = make_params(tmp, num_frames=nz, num_rows=ny, num_cols=nx, num_threads=n_nodes, **over)
    for node in range(n_nodes):
        ip, cfg = make_params(tmp, num_frames=nz, num_rows=ny, num_cols=nx, num_threads=n_nodes, **over)
        w = quiet(ref_writer.ReCoDeWriter, base, dark_data=dark, output_directory=tmp, input_params=ip,
                  mode="batch", validation_frame_gap=-1, node_id=node)
        quiet(w.start)
        quiet(w.run, frames)
        quiet(w.close)
    return cfg


def g3_g4():
    meta = {}
    cases = [
        # tag,     nz ny  nx  s     depth nodes overrides
        ("l1z12", 8, 40, 56, 0.10, 12, 3, dict()),
        ("l1z16", 5, 37, 53, 0.05, 16, 2, dict(source_bit_depth=16, target_bit_depth=16,
                                               calibration_threshold_epsilon=7)),
        ("l1ro16", 4, 16, 24, 0.20, 16, 2, dict(source_bit_depth=16, target_bit_depth=16,
                                                rc_operation_mode=0)),
        ("l3z", 4, 24, 40, 0.10, 12, 2, dict(reduction_level=3)),
    ]
    for tag, nz, ny, nx, s, depth, nodes, over in cases:
        tmp = tempfile.mkdtemp()
        dark, frames = synth_stack(2000 + len(meta), nz, ny, nx, s, depth)
        base = "g3_" + tag
        cfg = write_parts(tmp, base, dark, frames, nodes, **over)
        level = cfg["reduction_level"]
        names = []
        for node in range(nodes):
            fn = "%s.rc%d_part%03d" % (base, level, node)
            shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
            names.append(fn)
        quiet(ref_reader.merge_parts, tmp, "%s.rc%d" % (base, level), nodes)
        fn = "%s.rc%d" % (base, level)
        shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
        # decoded frames through the reference reader (L1 only: the reference cannot read L3)
        dec = None
        if level == 1:
            rd = ref_reader.ReCoDeReader(os.path.join(tmp, fn), is_intermediate=False)
            quiet(rd.open, print_header=False)
            dec = np.zeros_like(frames)
            for z in range(nz):
                fr = quiet(rd.get_frame, z)
                dec[z] = np.asarray(fr[z]["data"].todense()).astype(np.uint16)
            rd.close()
        np.savez_compressed(os.path.join(OUT, base + ".npz"), dark=dark, frames=frames,
                            cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array(list(cfg.values())),
                            n_nodes=nodes, decoded=dec if dec is not None else np.zeros(0))
        meta[tag] = names
        shutil.rmtree(tmp)
    print("g3_g4:", meta)


# --------------------------------------------------------------------------------------------
# G5: sparse expand through the reference's compiled extension
# --------------------------------------------------------------------------------------------
def g5():
    out = {}
    rng = np.random.default_rng(55)
    for tag, ny, nx, s, depth, level in [("a", 37, 53, 0.10, 12, 1), ("b", 40, 56, 0.05, 16, 1),
                                         ("c", 24, 40, 0.20, 9, 1), ("d", 24, 40, 0.10, 12, 3)]:
        binary = rng.random((ny, nx)) < s
        n = int(binary.sum())
        vals = rng.integers(1, 1 << depth, n).astype(np.uint16)
        bitmap = np.packbits(binary.ravel(), bitorder="little")
        packed = np.asarray(ref_writer._bit_pack(vals, depth), np.uint8)
        rd = ref_c.Reader()
        rd.create_buffers(ny, nx, depth)
        buf = bytearray(ny * nx * 3 * 8)
        got = rd.get_frame_sparse(level, bitmap.tobytes(), packed.tobytes(), buf)
        trip = np.frombuffer(bytes(buf), np.uint64, count=got * 3).reshape(got, 3).copy()
        out[f"g5_{tag}_shape"] = np.array([ny, nx, depth, level])
        out[f"g5_{tag}_bitmap"] = bitmap
        out[f"g5_{tag}_packed"] = packed
        out[f"g5_{tag}_vals"] = vals
        out[f"g5_{tag}_triplets"] = trip
    np.savez_compressed(os.path.join(OUT, "g5_expand.npz"), **out)
    print("g5:", len(out), "arrays")


def g6():
    """A version-0.1 header as the reference itself writes it: ReCoDeHeader(version=0.1).create(...).serialize(...)."""
    from pyrecode.params import InitParams
    tmp = tempfile.mkdtemp()
    try:
        ip, cfg = make_params(tmp, num_frames=7, num_rows=40, num_cols=56, num_threads=1, compression_scheme=0,
                              calibration_threshold_epsilon=3, frame_offset=2, source_bit_depth=12, target_bit_depth=12)
        init = InitParams("batch", tmp, image_filename="legacy_stack.bin", calibration_filename="legacy_dark.bin")
        h = ReCoDeHeader(version=0.1)
        quiet(h.create, init, ip, True)
        path = os.path.join(FILES, "g6_header_v01.bin")
        quiet(h.serialize, path)
        d = h.as_dict()
        keys = [k for k in d if not isinstance(d[k], (str, np.ndarray))]
        np.savez_compressed(os.path.join(OUT, "g6_header_v01.npz"), keys=np.array(keys), vals=np.array([int(d[k]) for k in keys], np.int64),
                            source_file_name=np.array(str(d["source_file_name"])), calibration_file_name=np.array(str(d["calibration_file_name"])))
        print("g6: %d bytes, %d scalar fields" % (os.path.getsize(path), len(keys)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# --------------------------------------------------------------------------------------------
# G7: mode='stream' - the writer is handed one chunk after the other (recode_writer.py:311-322,422-423): every chunk is split
# over the nodes by the same contiguous-block rule, frame ids run on from chunk to chunk (_chunk_offset), so a part file's ids
# are increasing but NOT contiguous.  Part files only: the reference's own merge misorders such parts.
# --------------------------------------------------------------------------------------------
def g7():
    chunks, ny, nx, nodes = (5, 4, 1, 6), 24, 40, 2
    dark, frames = synth_stack(2700, sum(chunks), ny, nx, 0.10, 12)
    tmp = tempfile.mkdtemp()
    try:
        base = "g7_stream"
        for node in range(nodes):
            ip, cfg = make_params(tmp, num_frames=1, num_rows=ny, num_cols=nx, num_threads=nodes)   # (<= every chunk: recode_writer.py:283-286)
            w = quiet(ref_writer.ReCoDeWriter, base, dark_data=dark, output_directory=tmp, input_params=ip,
                      mode="stream", validation_frame_gap=-1, node_id=node, run_name=base)      # (stream mode names its files after run_name, :193-194)
            quiet(w.start)
            at = 0
            for c in chunks:
                quiet(w.run, frames[at:at + c])
                at += c
            quiet(w.close)
        ids = []
        for node in range(nodes):
            fn = "%s.rc1_part%03d" % (base, node)
            shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
            rd = ref_reader.ReCoDeReader(os.path.join(tmp, fn), is_intermediate=True)
            quiet(rd.open, print_header=False)
            mine = []
            while True:
                f = quiet(rd.get_next_frame_raw, read_data=False)
                if not f:
                    break
                mine.append(int(list(f.keys())[0]))
            rd.close()
            ids.append(mine)
        np.savez_compressed(os.path.join(OUT, base + ".npz"), dark=dark, frames=frames, chunks=np.array(chunks),
                            cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array(list(cfg.values())), n_nodes=nodes,
                            ids_part0=np.array(ids[0]), ids_part1=np.array(ids[1]))
        print("g7:", ids)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# --------------------------------------------------------------------------------------------
# G8: validation frames (recode_writer.py:207-211,400-415): every validation_frame_gap-th frame (by absolute frame id) goes raw into
# <base>_partNNN_validation_frames.bin and its dose rate - 8-connected components of the binary map inside the central 128 x 128
# pixels / ROI pixels - into run_metrics['run_dose_rates'].  Clustered events, so that components are not single pixels.
# --------------------------------------------------------------------------------------------
def g8():
    nz, ny, nx, nodes, gap = 10, 150, 170, 2, 3
    rng = np.random.default_rng(2800)
    dark = rng.integers(3, 21, (ny, nx)).astype(np.uint16)
    frames = np.empty((nz, ny, nx), np.uint16)
    for z in range(nz):
        f = np.floor(rng.random((ny, nx)) * (dark + 1)).astype(np.uint16)
        for _ in range(int(rng.integers(40, 160))):                # blobs of 1..9 pixels, some touching only diagonally
            y, x = int(rng.integers(0, ny - 3)), int(rng.integers(0, nx - 3))
            shape = rng.random((3, 3)) < 0.45
            shape[1, 1] = True
            f[y:y + 3, x:x + 3] = np.where(shape, dark[y:y + 3, x:x + 3] + rng.integers(1, 3000, (3, 3)).astype(np.uint16), f[y:y + 3, x:x + 3])
        frames[z] = f
    tmp = tempfile.mkdtemp()
    try:
        base = "g8_valid"
        rates, vbytes = [], []
        for node in range(nodes):
            ip, cfg = make_params(tmp, num_frames=nz, num_rows=ny, num_cols=nx, num_threads=nodes)
            w = quiet(ref_writer.ReCoDeWriter, base, dark_data=dark, output_directory=tmp, input_params=ip,
                      mode="batch", validation_frame_gap=gap, node_id=node)
            quiet(w.start)
            m = quiet(w.run, frames)
            quiet(w.close)
            rates.append(np.array(m.get("run_dose_rates", []), np.float64))
            fn = "%s.rc1_part%03d" % (base, node)
            shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
            vbytes.append(np.frombuffer(open(os.path.join(tmp, "%s_part%03d_validation_frames.bin" % (base, node)), "rb").read(), np.uint8))
        np.savez_compressed(os.path.join(OUT, base + ".npz"), dark=dark, frames=frames, gap=gap,
                            cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array(list(cfg.values())), n_nodes=nodes,
                            rates_part0=rates[0], rates_part1=rates[1], validation_part0=vbytes[0], validation_part1=vbytes[1])
        print("g8:", [r.tolist() for r in rates], [v.size for v in vbytes])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# --------------------------------------------------------------------------------------------
# G9: the other host-only schemes of the standard library (recode_compressors.py:97-101: 4 = bz2, 5 = lzma) and a second zlib level:
# whole part files + merged file, as G3.
# --------------------------------------------------------------------------------------------
def g9():
    for tag, over in (("l1bz2", dict(compression_scheme=4, compression_level=3)), ("l1lzma", dict(compression_scheme=5, compression_level=2)),
                      ("l1z12_lvl9", dict(compression_scheme=0, compression_level=9)), ("l3ro", dict(reduction_level=3, rc_operation_mode=0))):
        nz, ny, nx, nodes = 5, 24, 40, 2
        tmp = tempfile.mkdtemp()
        dark, frames = synth_stack(2900 + len(tag), nz, ny, nx, 0.10, 12)
        base = "g9_" + tag
        cfg = write_parts(tmp, base, dark, frames, nodes, **over)
        level = cfg["reduction_level"]
        for node in range(nodes):
            fn = "%s.rc%d_part%03d" % (base, level, node)
            shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
        fn = "%s.rc%d" % (base, level)
        dec = np.zeros(0)
        readable = cfg["compression_scheme"] not in (4, 5)     # the reference's READER cannot open bz2 / lzma files: import_checks has no
        if readable:                                            # entry for them (recode_compressors.py:122-124, KeyError) - part files only
            quiet(ref_reader.merge_parts, tmp, fn, nodes)
            shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
        if level == 1 and readable:
            rd = ref_reader.ReCoDeReader(os.path.join(tmp, fn), is_intermediate=False)
            quiet(rd.open, print_header=False)
            dec = np.zeros_like(frames)
            for z in range(nz):
                dec[z] = np.asarray(quiet(rd.get_frame, z)[z]["data"].todense()).astype(np.uint16)
            rd.close()
        np.savez_compressed(os.path.join(OUT, base + ".npz"), dark=dark, frames=frames, cfg_keys=np.array(list(cfg.keys())),
                            cfg_vals=np.array(list(cfg.values())), n_nodes=nodes, decoded=dec)
        shutil.rmtree(tmp)
        print("g9:", tag)


# --------------------------------------------------------------------------------------------
# G10: sources of 8 bits and fewer - the reference's Python path takes whatever map_dtype yields (misc.py:41-49: uint8 for
# source_bit_depth <= 8); thr, frame > thr and the residuals are uint8 arithmetic, `.tobytes()` is one byte a value at d = 8
# (recode_writer.py:126-137,437-440,463-464), d = 6 goes through _bit_pack.  Frames handed over as uint16 are cast (:352-354).
# --------------------------------------------------------------------------------------------
def synth_stack_u8(seed, nz, ny, nx, sparsity, depth):
    rng = np.random.default_rng(seed)
    top = (1 << depth) - 1
    dark = rng.integers(2, 9, (ny, nx)).astype(np.uint8)
    frames = np.empty((nz, ny, nx), np.uint8)
    for z in range(nz):
        mask = rng.random((ny, nx)) < sparsity
        amp = rng.integers(1, top - 12, (ny, nx)).astype(np.uint8)
        below = np.floor(rng.random((ny, nx)) * (dark + 1)).astype(np.uint8)
        frames[z] = np.where(mask, dark + amp, below)
    frames[0, 0, 0] = top                      # the largest value the depth holds
    frames[nz - 1, ny - 1, nx - 1] = top
    return dark, frames


def g10():
    cases = [
        # tag      nz ny  nx  s     depth nodes gap  handed over as   overrides
        ("u8d8",   7, 37, 53, 0.08, 8, 2, -1, np.uint8, dict(calibration_threshold_epsilon=2)),
        ("u8d6",   5, 24, 40, 0.12, 6, 2, -1, np.uint8, dict()),
        ("u8d8v",  6, 150, 170, 0.03, 8, 2, 2, np.uint8, dict()),            # validation frames: the side file holds uint8 frames
        ("u8cast", 4, 24, 40, 0.10, 8, 2, -1, np.uint16, dict()),            # frames handed over as uint16: the writer casts them
    ]
    for tag, nz, ny, nx, s, depth, nodes, gap, given, over in cases:
        tmp = tempfile.mkdtemp()
        dark, frames = synth_stack_u8(3100 + len(tag) + depth, nz, ny, nx, s, depth)
        base = "g10_" + tag
        over = dict(over, source_bit_depth=depth, target_bit_depth=depth)
        rates, vbytes = [], []
        for node in range(nodes):
            ip, cfg = make_params(tmp, num_frames=nz, num_rows=ny, num_cols=nx, num_threads=nodes, **over)
            w = quiet(ref_writer.ReCoDeWriter, base, dark_data=dark, output_directory=tmp, input_params=ip, mode="batch",
                      validation_frame_gap=gap, node_id=node)
            quiet(w.start)
            m = quiet(w.run, frames.astype(given))
            quiet(w.close)
            rates.append(np.asarray(m.get("run_dose_rates", []), np.float64))
            fn = "%s.rc1_part%03d" % (base, node)
            shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
            if gap > 0:
                vbytes.append(np.fromfile(os.path.join(tmp, "%s_part%03d_validation_frames.bin" % (base, node)), np.uint8))
        fn = base + ".rc1"
        quiet(ref_reader.merge_parts, tmp, fn, nodes)
        shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
        rd = ref_reader.ReCoDeReader(os.path.join(tmp, fn), is_intermediate=False)
        quiet(rd.open, print_header=False)
        dec = np.zeros_like(frames)
        dts = set()
        for z in range(nz):
            m = quiet(rd.get_frame, z)[z]["data"]
            dts.add(str(m.dtype))
            dec[z] = np.asarray(m.todense())
        rd.close()
        extra = {}
        for i, r in enumerate(rates):
            extra["rates%d" % i] = r
        for i, v in enumerate(vbytes):
            extra["vframes%d" % i] = v
        np.savez_compressed(os.path.join(OUT, base + ".npz"), dark=dark, frames=frames, cfg_keys=np.array(list(cfg.keys())),
                            cfg_vals=np.array(list(cfg.values())), n_nodes=nodes, decoded=dec, gap=gap, given=str(np.dtype(given)),
                            decoded_dtype=",".join(sorted(dts)), **extra)
        shutil.rmtree(tmp)
        print("g10:", tag, "decoded dtype", dts, "decoded == where(frame > thr, frame - thr, 0):",
              bool(np.array_equal(dec, np.where(frames > (dark + cfg["calibration_threshold_epsilon"]).astype(np.uint8),
                                                  frames - (dark + cfg["calibration_threshold_epsilon"]).astype(np.uint8), 0))))


# --------------------------------------------------------------------------------------------
# G11: sources beyond 16 bits - map_dtype yields uint32 (misc.py:41-49).  d = 20 goes through _bit_pack (20-bit fields), d = 32 and
# d = 24 through `.tobytes()` (recode_writer.py:463-464: FOUR bytes a value for both - the reader then takes 24-bit fields out of the
# 32-bit values of a d = 24 file and decodes something else than was written: the files are what is pinned, and what each reader returns).
# --------------------------------------------------------------------------------------------
def synth_stack_u32(seed, nz, ny, nx, sparsity, depth):
    rng = np.random.default_rng(seed)
    top = (1 << depth) - 1
    dark = rng.integers(1000, 70000, (ny, nx)).astype(np.uint32)
    frames = np.empty((nz, ny, nx), np.uint32)
    for z in range(nz):
        mask = rng.random((ny, nx)) < sparsity
        amp = rng.integers(1, top - 70000, (ny, nx), dtype=np.int64).astype(np.uint32)
        below = np.floor(rng.random((ny, nx)) * (dark + 1.0)).astype(np.uint32)
        frames[z] = np.where(mask, dark + amp, below)
    frames[0, 0, 0] = top
    frames[nz - 1, ny - 1, nx - 1] = top
    frames[0, 0, 1] = dark[0, 1] + 1              # the smallest residual
    frames[0, 0, 2] = dark[0, 2] + 65536          # a residual whose low 16 bits are zero
    return dark, frames


def g11():
    cases = [
        # tag       nz ny  nx  s     depth nodes overrides
        ("u32d20",  6, 37, 53, 0.08, 20, 2, dict(calibration_threshold_epsilon=3)),
        ("u32d32",  4, 24, 40, 0.10, 32, 2, dict()),
        ("u32d24",  4, 24, 40, 0.10, 24, 2, dict()),
        ("u32d17",  4, 40, 56, 0.06, 17, 2, dict()),                             # the narrowest uint32 depth
        ("u32d20v", 6, 150, 170, 0.03, 20, 2, dict(_gap=2)),                     # validation frames: the side file holds uint32 frames
    ]
    for tag, nz, ny, nx, s, depth, nodes, over in cases:
        tmp = tempfile.mkdtemp()
        dark, frames = synth_stack_u32(3300 + depth, nz, ny, nx, s, depth)
        base = "g11_" + tag
        gap = over.pop("_gap", -1)
        over = dict(over, source_bit_depth=depth, target_bit_depth=depth)
        ok = True
        rates, vbytes = [], []
        try:
            for node in range(nodes):
                ip, cfg = make_params(tmp, num_frames=nz, num_rows=ny, num_cols=nx, num_threads=nodes, **over)
                w = quiet(ref_writer.ReCoDeWriter, base, dark_data=dark, output_directory=tmp, input_params=ip, mode="batch",
                          validation_frame_gap=gap, node_id=node)
                quiet(w.start)
                m = quiet(w.run, frames)
                quiet(w.close)
                rates.append(np.asarray(m.get("run_dose_rates", []), np.float64))
                if gap > 0:
                    vbytes.append(np.fromfile(os.path.join(tmp, "%s_part%03d_validation_frames.bin" % (base, node)), np.uint8))
        except Exception as e:
            print("g11:", tag, "the reference's writer raised", repr(e))
            shutil.rmtree(tmp)
            continue
        for node in range(nodes):
            fn = "%s.rc1_part%03d" % (base, node)
            shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
        fn = base + ".rc1"
        quiet(ref_reader.merge_parts, tmp, fn, nodes)
        shutil.copy(os.path.join(tmp, fn), os.path.join(FILES, fn))
        dec, dts = np.zeros((nz, ny, nx), np.uint64), set()
        try:
            rd = ref_reader.ReCoDeReader(os.path.join(tmp, fn), is_intermediate=False)
            quiet(rd.open, print_header=False)
            for z in range(nz):
                m = quiet(rd.get_frame, z)[z]["data"]
                dts.add(str(m.dtype))
                dec[z] = np.asarray(m.todense()).astype(np.uint64)
            rd.close()
        except Exception as e:
            print("g11:", tag, "the reference's reader raised", repr(e))
            ok = False
        thr = (dark + np.uint32(cfg["calibration_threshold_epsilon"])).astype(np.uint32)
        want = np.where(frames > thr, frames - thr, 0).astype(np.uint64)
        np.savez_compressed(os.path.join(OUT, base + ".npz"), dark=dark, frames=frames, cfg_keys=np.array(list(cfg.keys())),
                            cfg_vals=np.array(list(cfg.values())), n_nodes=nodes, decoded=dec if ok else np.zeros(0, np.uint64),
                            decoded_dtype=",".join(sorted(dts)), gap=gap, **{"rates%d" % i: r for i, r in enumerate(rates)},
                            **{"vframes%d" % i: v for i, v in enumerate(vbytes)})
        shutil.rmtree(tmp)
        print("g11:", tag, "decoded dtype", dts, "reader ok", ok, "decoded == where(frame > thr, frame - thr, 0):", bool(ok and np.array_equal(dec, want)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g11":
        g11()
    elif len(sys.argv) > 1 and sys.argv[1] == "g10":
        g10()
    elif len(sys.argv) > 1 and sys.argv[1] == "g9":
        g9()
    elif len(sys.argv) > 1 and sys.argv[1] == "g8":
        g8()
    elif len(sys.argv) > 1 and sys.argv[1] == "g6":
        g6()
    elif len(sys.argv) > 1 and sys.argv[1] == "g7":
        g7()
    else:
        g1_g2()
        g3_g4()
        g5()
        g6()
        g7()
        g8()
        g9()
        g10()
        g11()
