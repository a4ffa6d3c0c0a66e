#!/usr/bin/env python3
"""bench.py - throughput of the per-frame reduce -> bit-pack -> compress hot path on N MI355X GPUs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).  With no RANK in the environment and
N > 1 this process starts its own N ranks - a CHILD `python -m torch.distributed.run --nproc-per-node N bench.py ...`, before it has
touched the GPU - relays rank 0's line and exits non-zero if any rank fails; launched by torch.distributed.run itself it is a rank
(one rank per GPU, RCCL).

  step      one pass of the hot path over one batch of B synthetic frames per GPU, inputs resident in HBM:
            rc_reduce_compress_batch_async = reduce kernel (every device codec's block encoder fused in) -> scans -> record
            layout -> k_gather (zstd: the serial FSE chain and the residuals' Huffman stage are kernels of their own in between),
            records + offsets + metadata left in HBM; for N > 1 each step also issues the path's one exchange step, the
            RCCL all-gather of the per-frame metadata (12 B / frame), on a side stream so that it overlaps
            the next step (metadata rows double-buffered; all of them are complete inside the timed region):
            pyrecode_amd/parallel.py::ShardedStepLoop.
  workload  BASELINE.json configs[1]: 4096x4096 uint16, 1 % sparsity, L1 + LZ4 (d = 16 primary; --depth 12 secondary);
            --config 1..5 = the other BASELINE configurations, --clustered = detector-like events.
  value     frames/s, whole job (all ranks' frames / max-over-ranks time); gb_per_s_in = value * 2*nx*ny.
  verified  every rank decodes TWO records of every distinct batch of its stack - one inside the batch and its last - with the stock
            library / the oracle's decoders and compares them with the oracle's reduce of those frames (level 2: scipy.ndimage.label);
            gather_verified: every rank's block of the metadata table the last step gathered, on every rank.
  roofline  dominant kernel = the reduce kernel (k_reduce_tiles): algorithmic bytes per launch = B * 2*nx*ny
            (one read of the uint16 frames) / its mean duration, measured with HIP events on the ctx's
            stream inside the timed region - around every --kernel-events-every-th launch (default 4: the events are packets of
            their own between two reduce kernels); peak = 8000 GB/s (MI355X_MICROARCH.md).
  cpu_baseline  the oracle's C restatement (+ stock liblz4 for the LZ4 stage) on the box's host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0


# BASELINE.json `configs`, as bench.py arguments (per-GPU workload; config 3 / 5 are the 8-GPU jobs: --gpus 8 shards them)
CONFIGS = {
    1: dict(ny=512, nx=512, sparsity_ppm=145000, depth=12, scheme=0, level=1, batch=9, stack=9),   # minimal_read_write_test: zlib on the host
    2: dict(ny=4096, nx=4096, sparsity_ppm=10000, depth=16, scheme=2, level=1),                     # the headline (defaults)
    3: dict(ny=4096, nx=4096, sparsity_ppm=10000, depth=16, scheme=1, level=1),
    4: dict(ny=4096, nx=4096, sparsity_ppm=1000, depth=16, scheme=8, level=2),
    5: dict(ny=8184, nx=11520, sparsity_ppm=50000, depth=12, scheme=1, level=1, batch=32, stack=64),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="BASELINE.json configs[k-1] (1..5) as defaults for the workload flags below; 0 = the headline, configs[1]")
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--stack", type=int, default=256, help="distinct frames resident per GPU")
    ap.add_argument("--ny", type=int, default=4096)
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--sparsity-ppm", type=int, default=10000)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--clustered", action="store_true", help="detector-like events: clusters of 1..6 pixels; --sparsity-ppm then counts SEEDS per million pixels (11000 = ~4.3 %% set pixels, the real acquisition the reference's notebook records)")
    ap.add_argument("--scheme", type=int, default=2, help="2 = LZ4 (headline), 1 = zstd, 8 = blosc-lz4, 0 = reduce-only pieces")
    ap.add_argument("--device-zlib", action="store_true", help="--scheme 0: both streams of a record made by the device's DEFLATE encoder (RC_SCHEME_ZLIB_DEVICE: zlib streams that "
                    "stdlib zlib expands to the oracle's bytes) instead of the reduce-only pieces for the host's zlib.compress")
    ap.add_argument("--level", type=int, default=1, help="reduction level: 1 (headline), 2 = summary statistics, 3 = bitmap only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--source-bytes", type=int, default=2, choices=[1, 2, 4], help="bytes per source pixel: 2 = uint16 frames (every BASELINE configuration), 1 = uint8 frames (source_bit_depth <= 8: the reference's map_dtype, misc.py:41-49; implies --depth 8 unless given lower), 4 = uint32 frames (> 16 bits; implies --depth 20 unless given above 16); both without the CPU baseline / ingest legs")
    ap.add_argument("--read", action="store_true", help="measure the READER instead: stored frames -> device decode of both streams -> sparse expand (rc_expand_frames)")
    ap.add_argument("--blob-on-device", action="store_true", help="--read: the stored frames' bytes already sit in device memory (the decoders without the link)")
    ap.add_argument("--kernel-events-every", type=int, default=4, help="HIP events around every k-th reduce-kernel launch of the timed region (roofline.kernel_ms is their mean; "
                    "1: every launch).  A timing event is a packet of its own between two reduce kernels: with events around every launch the step is 1.3 %% longer "
                    "(same box, profiles/r05_exp12_level2_listed_words_events.log)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the extra ingest-inclusive measurement (host frames -> part file)")
    ap.add_argument("--ingest-frames", type=int, default=512, help="frames per pass of the ingest-inclusive measurement (capped at 16 GiB of host memory)")
    ap.add_argument("--no-pipeline", action="store_true", help="plain stream order: a batch's reduce kernel waits for the previous batch's records")
    ap.add_argument("--clevel", type=int, default=1, help="compression_level: 0 = the fast device encoders, >= 1 = the modelled zstd encoder")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="the K-step timed region is repeated until this much time has been measured (>= 3 repeats); the median repeat is reported")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (the product); gloo only to rehearse on a box with fewer GPUs than ranks")
    ap.add_argument("--shared-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --dist-backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--gather-every", type=int, default=0, help="steps per metadata all-gather: 0 = ONE gather per timed region (default - BASELINE: 'RCCL only for the final merged-index gather'), 1 = every step (the form of rounds 1-3), K = every K steps")
    ap.add_argument("--no-collective", action="store_true", help="rehearsal: the ranks only meet at the fences (barrier, max-over-ranks time); no metadata gather - isolates the HOST side of N step loops from the rehearsal backend's host copies")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin each rank to its share of the usable CPUs")
    pre, _ = ap.parse_known_args(argv)
    if pre.config:
        ap.set_defaults(**CONFIGS[pre.config])
    a = ap.parse_args(argv)
    if a.source_bytes == 1:
        a.depth = min(a.depth, 8)
        a.no_cpu_baseline = a.no_ingest = True
    if a.source_bytes == 4:
        a.depth = a.depth if a.depth > 16 else 20
        a.no_cpu_baseline = a.no_ingest = True
    if a.shared_gpu and a.dist_backend == "nccl":
        ap.error("--shared-gpu needs --dist-backend gloo")
    return a


def cpu_baseline(frames_h, thr_h, depth, scheme, zlib_level=None):
    """Oracle (CPU restatement, oracle/recode_oracle.c) + stock liblz4 on the host cores; bounded to ~15 core-seconds.
    zlib_level (scheme 0 with the device encoder): the reference's own compress stage - zlib.compress(data, level) on both streams
    (recode_compressors.py:84-85; the call releases the GIL)."""
    import ctypes as C
    import ctypes.util
    import threading
    import zlib
    from oracle import oracle as orc
    orc.lib()
    lz4 = None
    if scheme == 2:
        name = ctypes.util.find_library("lz4")
        if name:
            lz4 = C.CDLL(name)
            lz4.LZ4F_compressFrameBound.restype = C.c_size_t
            lz4.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
            lz4.LZ4F_compressFrame.restype = C.c_size_t
            lz4.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    n_pix = frames_h.shape[1]
    thr_flat = np.ascontiguousarray(thr_h.ravel())

    def one(frame, scratch):
        bitmap, packed, nnz = orc.reduce_frame_l1(frame, thr_flat, depth)
        if lz4 is not None:
            for src in (bitmap, packed):
                n = lz4.LZ4F_compressFrame(scratch.ctypes.data, scratch.size, src.ctypes.data, src.size, None)
                assert n > 0
        if zlib_level is not None:
            for src in (bitmap, packed):
                zlib.compress(src, zlib_level)
        return nnz

    bound = int(lz4.LZ4F_compressFrameBound(n_pix // 8 + 64, None)) + n_pix * 2 if lz4 is not None else 16
    t0 = time.perf_counter()
    one(frames_h[0], np.empty(bound, np.uint8))
    t_one = time.perf_counter() - t0
    from pyrecode_amd.misc import effective_cpus
    threads, visible = effective_cpus()   # every core this process may REALLY use: the affinity mask capped by the cgroup's CPU quota
    nfr = frames_h.shape[0]
    budget_s = 12.0                      # wall-clock bound: every worker runs until the deadline and counts its frames
    done = [0] * threads

    def work(i):   # worker i takes frames i, i + threads, ... (mod nfr) until the deadline
        scratch = np.empty(bound, np.uint8)
        j, k = i, 0
        while time.perf_counter() < deadline:
            one(frames_h[j % nfr], scratch)
            j += threads
            k += 1
        done[i] = k

    # single core first (short), then all workers
    t0 = time.perf_counter()
    scratch = np.empty(bound, np.uint8)
    n1 = min(8, frames_h.shape[0])
    for f in frames_h[:n1]:
        one(f, scratch)
    single = n1 / (time.perf_counter() - t0)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    deadline = t0 + budget_s
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(total / dt, 2), "unit": "frames/s", "cores": threads, "cores_visible": visible, "cpu_model": model, "kind": "port",
        "single_core_value": round(single, 2),
        "sample": "%d frame passes over %d distinct synthetic frames of the GPU stack (%.1f s wall), oracle C reduce+pack%s" % (
            total, nfr, dt, " + liblz4 LZ4F_compressFrame on bitmap and pixvals" if lz4 is not None else
            (" + zlib.compress(level %d) on bitmap and pixvals (stdlib zlib: the reference's own call)" % zlib_level if zlib_level is not None else
             (" (no compress stage: liblz4 not found)" if scheme == 2 else " (reduce-only)"))),
    }


def ingest_inclusive(stack, dark, a, nframes=512, validation_frame_gap=-1, passes=3, data=None, device_zlib=None):
    """Extra, NOT `value`: ReCoDeWriter.run on frames that start in host memory, records appended to a part file on tmpfs
    (the reference's whole writer loop, recode_writer.py:292-428): staging copy + link + kernels + records back + file append.
    >= 512 frames per pass where they fit 16 GiB of host memory (0.4 s at 4096^2: long enough to tell a stall from noise), one warm-up
    pass (staging buffers, model) and `passes` timed ones: median, min and max are reported."""
    import shutil
    import tempfile
    from pyrecode_amd.params import InputParams
    from pyrecode_amd.recode_writer import ReCoDeWriter
    frame_bytes = a.ny * a.nx * 2
    n = int(max(min(nframes, (16 << 30) // frame_bytes), min(stack.shape[0], 8)))
    if data is None:
        src = stack[:min(n, stack.shape[0])].cpu().numpy().view(np.uint16).reshape(-1, a.ny, a.nx)
        data = src if src.shape[0] >= n else np.concatenate([src] * (-(-n // src.shape[0])))[:n]   # (the stack's frames again: other ids)
    n = data.shape[0]
    dark_h = dark.cpu().numpy().view(np.uint16).reshape(a.ny, a.nx)
    ip = InputParams()
    ip._param_map.update(dict(reduction_level=a.level, rc_operation_mode=1, calibration_threshold_epsilon=0, target_bit_depth=a.depth,
                              source_bit_depth=a.depth, num_cols=a.nx, num_rows=a.ny, num_frames=n, frame_offset=0,
                              num_calibration_frames=1, calibration_frame_offset=0, keep_part_files=1, num_threads=1, l2_statistics=0,
                              l4_centroiding=0, compression_scheme=a.scheme, compression_level=a.clevel, source_file_type=0,
                              source_header_length=0, keep_calibration_data=0, calibration_file_type=0, source_data_type=0,
                              target_data_type=0))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    out_dir = tempfile.mkdtemp(dir=base)
    try:
        times = []
        for k in range(passes + 1):   # first pass warms the staging buffers and the model
            w = ReCoDeWriter("bench_stack.bin", dark_data=dark_h, output_directory=out_dir, input_params=ip, mode="batch", node_id=0,
                             batch_size=min(32, a.batch), validation_frame_gap=validation_frame_gap,
                             device_zlib=(a.device_zlib if device_zlib is None else device_zlib))
            w.start()
            t0 = time.perf_counter()
            w.run(data)
            dt = time.perf_counter() - t0
            w.close()
            if k:
                times.append(dt)
        times.sort()
        med = times[len(times) // 2]
        part = os.path.join(out_dir, "bench_stack.rc%d_part000" % a.level)
        size = os.path.getsize(part)
        res = {"frames_per_s": round(n / med, 1), "gb_per_s_in": round(n * frame_bytes / med / 1e9, 2), "frames": n, "passes": len(times),
               "frames_per_s_min_max": [round(n / times[-1], 1), round(n / times[0], 1)], "seconds_per_pass": round(med, 3),
               "part_file_bytes": size, "what": "ReCoDeWriter.run: host frames -> page-locked staging -> GPU -> records -> part file on %s; median of %d passes" % (base or "tmp", len(times))}
        if validation_frame_gap <= 0 and a.level in (1, 3):
            res["read_back"] = read_back(part, n, int(sum(int((data[lo:lo + 64] > dark_h[None]).sum()) for lo in range(0, n, 64))))
        return res, data
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def read_back(part_file, n, nnz_expected):
    """Extra, NOT `value`: the part file just written, read back through ReCoDeReader with the triplets delivered to the HOST - the batched
    iterator (24-byte (row, col, value) rows: bound by the link) and the reference's frame-at-a-time call (get_next_frame, which reads
    ahead through the batched reader once sequential and wraps every frame as a scipy COO matrix, reference recode_reader.py:188-221)."""
    from pyrecode_amd.recode_reader import ReCoDeReader
    out = {}
    rd = ReCoDeReader(part_file, is_intermediate=True)
    rd.open(print_header=False)
    try:
        for _ in range(2):
            t0 = time.perf_counter()
            got = 0
            for _, prefix, _ in rd.iter_frames_triplets(batch=32):
                got += int(prefix[-1])
            dt = time.perf_counter() - t0
        out["iter_frames_triplets_frames_per_s"] = round(n / dt, 1)
        out["verified"] = bool(got == nnz_expected)
        for _ in range(2):                       # the same batches as the COO arrays (int32 rows, int32 columns, uint16 values: 10 bytes a set pixel)
            t0 = time.perf_counter()
            got = 0
            for _, prefix, (rows, _, _) in rd.iter_frames_triplets(batch=32, coo=True):
                got += rows.shape[0]
            dt = time.perf_counter() - t0
        out["iter_frames_coo_frames_per_s"] = round(n / dt, 1)
        out["verified"] = bool(out["verified"] and got == nnz_expected)
        t0 = time.perf_counter()
        k = nnz = 0
        while True:
            f = rd.get_next_frame()
            if f is None:
                break
            (_, body), = f.items()
            nnz += body["data"].nnz
            k += 1
        dt = time.perf_counter() - t0
        out["get_next_frame_frames_per_s"] = round(k / dt, 1)
        out["verified"] = bool(out["verified"] and k == n and nnz == nnz_expected)
        out["path"] = rd.last_batch_path
    finally:
        rd.close()
    return out


def bench_read(a, emit=True):
    """Reader line: a step = one batch of B stored frames through the batched reader (host blobs in, triplets left in device memory),
    streamed with two batches in flight (rc_expand_frames_submit / _wait); the synchronous one-call form is reported next to it.
    Roofline of the path: the decoded streams are read once by the expand kernels and 24 bytes are written per set pixel
    (row, col, value as uint64, the reference's triplet format, pyrecode.cpp:95-119): algorithmic bytes per frame =
    nb + n_packed + 24 * nnz."""
    import torch
    from pyrecode_amd import _lib as hip
    L = hip.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N, B = a.ny * a.nx, a.batch
    dark = torch.empty(N, dtype=torch.int16, device=dev)
    stack = torch.empty((B, N), dtype=torch.int16, device=dev)
    hip.check(L.rc_synth_dark(0, 20261003, N, dark.data_ptr()))
    hip.check(L.rc_synth_frames(0, 20261003, 0, B, N, a.sparsity_ppm, dark.data_ptr(), stack.data_ptr()))
    ctx = hip.ReduceContext(a.nx, a.ny, a.depth, a.level, 1, a.scheme, a.clevel, 0, max_batch=B)
    ctx.set_dark(dark.data_ptr(), 0)
    out, rec, md = ctx.reduce_compress_batch(stack.cpu().numpy().view(np.uint16).reshape(B, a.ny, a.nx), first_frame_id=0)
    ctx.close()
    hdr = 16 if a.level == 1 else 8
    blobs, sizes = [], np.zeros((B, 3), np.uint32)
    for z in range(B):
        blobs.append(out[int(rec[z]) + hdr:int(rec[z + 1])])
        sizes[z] = md[z] if a.level == 1 else (md[z][0], 0, 0)
    pinned = hip.PinnedBuffer(sum(b.size for b in blobs))   # what a reader holds: the file's bytes in page-locked memory
    blob = pinned.array
    np.concatenate(blobs, out=blob)
    prefix = np.zeros(B + 1, np.uint64)
    blob_arg = hip.ptr(blob)
    if a.blob_on_device:   # the stored frames' bytes already in device memory: the decoders without the link
        blob_dev = torch.zeros(blob.size + 4096, dtype=torch.uint8, device=dev)
        blob_dev[:blob.size].copy_(torch.from_numpy(blob))
        blob_arg = blob_dev.data_ptr()
    args = (a.nx, a.ny, a.depth, a.level, 1, a.scheme, blob_arg, hip.ptr(sizes), B)
    hip.check(L.rc_expand_frames(*args, hip.ptr(prefix), None, 0), "rc_expand_frames")
    nnz = int(prefix[B])
    trip = torch.empty((max(nnz, 1), 3), dtype=torch.int64, device=dev)
    for _ in range(max(a.warmup, 2)):
        hip.check(L.rc_expand_frames(*args, hip.ptr(prefix), trip.data_ptr(), nnz), "rc_expand_frames")
    times = []
    while len(times) < 3 or (sum(times) < a.min_seconds and len(times) < 200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            hip.check(L.rc_expand_frames(*args, hip.ptr(prefix), trip.data_ptr(), nnz), "rc_expand_frames")
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    # the streaming form: two batches in flight (rc_expand_frames_submit / _wait), the host walk + copy-in of one under the decode of the other
    trip2 = [trip, torch.empty_like(trip)]
    pargs = (a.nx, a.ny, a.depth, a.level, 1, a.scheme, blob_arg, hip.ptr(sizes), B)
    ptimes = []
    while len(ptimes) < 3 or (sum(ptimes) < a.min_seconds and len(ptimes) < 200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            hip.check(L.rc_expand_frames_submit(k & 1, *pargs, trip2[k & 1].data_ptr(), nnz), "rc_expand_frames_submit")
            if k:
                hip.check(L.rc_expand_frames_wait((k - 1) & 1, hip.ptr(prefix)), "rc_expand_frames_wait")
        hip.check(L.rc_expand_frames_wait((a.steps - 1) & 1, hip.ptr(prefix)), "rc_expand_frames_wait")
        torch.cuda.synchronize()
        ptimes.append(time.perf_counter() - t0)
    pdt = sorted(ptimes)[len(ptimes) // 2]
    assert torch.equal(trip2[0], trip2[1])
    # the same stream of batches with the output in the COO layout (int32 rows | int32 columns | uint16 values: rc_expand_frames_coo_submit)
    coo_line = None
    if a.depth <= 16:
        coo2 = [torch.empty(10 * max(nnz, 1) + 16, dtype=torch.uint8, device=dev) for _ in range(2)]
        ctimes = []
        while len(ctimes) < 3 or (sum(ctimes) < a.min_seconds / 2 and len(ctimes) < 100):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(a.steps):
                hip.check(L.rc_expand_frames_coo_submit(k & 1, *pargs, coo2[k & 1].data_ptr(), nnz), "rc_expand_frames_coo_submit")
                if k:
                    hip.check(L.rc_expand_frames_wait((k - 1) & 1, hip.ptr(prefix)), "rc_expand_frames_wait")
            hip.check(L.rc_expand_frames_wait((a.steps - 1) & 1, hip.ptr(prefix)), "rc_expand_frames_wait")
            torch.cuda.synchronize()
            ctimes.append(time.perf_counter() - t0)
        cdt = sorted(ctimes)[len(ctimes) // 2]
        host = coo2[0].cpu().numpy()
        t_all = trip.cpu().numpy().view(np.uint64)
        same = (np.array_equal(host[:4 * nnz].view(np.int32), t_all[:nnz, 0].astype(np.int32))
                and np.array_equal(host[4 * nnz:8 * nnz].view(np.int32), t_all[:nnz, 1].astype(np.int32))
                and np.array_equal(host[8 * nnz:10 * nnz].view(np.uint16), t_all[:nnz, 2].astype(np.uint16)))
        coo_line = {"frames_per_s": round(B * a.steps / cdt, 1), "ms_per_step": round(cdt / a.steps * 1e3, 4), "equals_the_triplets": bool(same),
                    "what": "the streaming form with the output as int32 rows | int32 columns | uint16 values (10 instead of 24 bytes per set pixel)"}
    # verification: frame B//2 against the oracle's expand of the oracle's reduce
    from oracle import oracle as orc
    z = B // 2
    frame = stack[z].cpu().numpy().view(np.uint16)
    thr_h = dark.cpu().numpy().view(np.uint16)
    bitmap, packed, _ = orc.reduce_frame_l1(frame, thr_h, a.depth)
    want = orc.unpack_frame_sparse(a.nx, a.ny, a.depth, bitmap, packed, a.level)
    got = trip[int(prefix[z]):int(prefix[z + 1])].cpu().numpy().view(np.uint64)
    fps_call, fps = B * a.steps / dt, B * a.steps / pdt
    dt_call, dt = dt, pdt
    alg = (N // 8) * B + int(sizes[:, 2].sum()) + 24 * nnz
    line = {
        "metric": "reader: frames/sec, stored frames -> decode both streams -> (row, col, value) triplets in device memory",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
        "one_call_at_a_time": {"frames_per_s": round(fps_call, 1), "ms_per_call": round(dt_call / a.steps * 1e3, 4),
                               "what": "rc_expand_frames, synchronous; value is the streaming form, two batches in flight (rc_expand_frames_submit / _wait)"},
        "repeats": len(times), "higher_is_better": True, "dtype": "u8/u64", "data": "synthetic", "verified": bool(np.array_equal(got, want)),
        "config": {"workload": "%dx%d, %.2f%% sparsity, L%d, scheme %d (clevel %d), depth %d, %d frames per call; input = the records' data blobs in %s memory (%.0f B/frame)"
                               % (a.ny, a.nx, a.sparsity_ppm / 1e4, a.level, a.scheme, a.clevel, a.depth, B, "DEVICE" if a.blob_on_device else "host", blob.size / B)},
        "roofline": {"bound": "hbm", "achieved": round(alg * a.steps / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg * a.steps / dt / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes_per_call": alg, "note": "whole call incl. host-side block indexing and the copy-in of the compressed blobs over the link (%.0f B/frame: the link alone allows about %.0f k frames/s); the decoders are serial chains per block (latency bound), not bandwidth bound" % (blob.size / B, 57e9 / (blob.size / B) / 1e3)},
        "nnz_per_frame": round(nnz / B, 1)}
    if coo_line is not None:
        line["coo_layout"] = coo_line
    if emit:
        print(json.dumps(line), flush=True)
    return line


def launch_children(a, argv):
    """`python bench.py --gpus N` with no rank in the environment: start the N ranks ourselves, as the reference's server starts
    its N writer processes (recode_server.py:350-363).  This process has not touched the GPU (no torch import, no HIP call) and
    never does: the ranks are CHILDREN (`python -m torch.distributed.run`), rank 0's one JSON line is relayed, and a failure
    of any rank is this process's failure."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, RC_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")              # N step loops share the host: no rank spins up a thread pool of its own
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank child job failed (exit code %d); its ranks' messages are above\n" % (a.gpus, rc))
        return rc
    if line is None:
        sys.stderr.write("bench.py: the child job printed no result line\n")
        return 1
    print(line, flush=True)
    return 0


def verify_record(a, r, frame, thr_h, frame_id):
    """One record against the oracle (CPU restatement): the stock decoder / the oracle's decoders must expand both streams to
    the frame's exact binary map and value list.  Level 2 (no runnable reference): scipy.ndimage.label + numpy."""
    import struct
    from oracle import oracle as orc
    if frame.dtype == np.uint32:      # sources beyond 16 bits: the oracle's numpy restatement (pinned on fixture G11)
        binary = frame > thr_h
        bitmap, nnz = np.packbits(binary, bitorder="little"), int(binary.sum())
        packed = orc.bit_pack32((frame[binary] - thr_h[binary]).astype(np.uint32), a.depth)
    else:
        bitmap, packed, nnz = orc.reduce_frame_l1(frame, thr_h, a.depth)
    if a.level == 2:
        import scipy.ndimage as nd
        f2, t2 = frame.reshape(a.ny, a.nx), thr_h.reshape(a.ny, a.nx)
        labels, n = nd.label(f2 > t2, structure=np.ones((3, 3), int))          # recode_writer.py:166,443
        vals = nd.maximum(f2.astype(np.int64), labels, np.arange(1, n + 1)) if n else np.zeros(0)   # l2_statistics 0: max
        packed = orc.bit_pack((np.asarray(vals, np.int64) & ((1 << a.depth) - 1)).astype(np.uint16), a.depth)   # (a sum wraps at d bits, as the reference's cast + pack would)
    bitmap, packed = bitmap.tobytes(), packed.tobytes()
    if a.scheme == 2:
        dec = lambda b, n: orc.lz4f_decode(b, n + 8)
    elif a.scheme == 1:
        from pyrecode_amd.recode_compressors import _zstd_host_decompress
        dec = lambda b, n: _zstd_host_decompress(b)
    elif a.scheme == 8:
        dec = lambda b, n: orc.blosc1_decode(b)
    elif a.scheme == 0 and a.device_zlib:
        import zlib
        dec = lambda b, n: zlib.decompress(b)      # stdlib zlib: the reference reader's own call (recode_compressors.py:43)
    else:
        dec = None
    if a.level in (1, 2):
        if dec is None:   # reduce-only pieces (zlib & co. are the host library's call, as in the reference)
            fid, npk = struct.unpack_from("<II", r, 0)
            return fid == frame_id and r[8:] == bitmap + packed
        fid, cb, cp, npk = struct.unpack_from("<IIII", r, 0)
        return (fid == frame_id and npk == len(packed) and len(r) == 16 + cb + cp and dec(r[16:16 + cb], len(bitmap)) == bitmap
                and dec(r[16 + cb:], npk) == packed)
    if a.level == 3:
        if dec is None:
            return struct.unpack_from("<I", r, 0)[0] == frame_id and r[4:] == bitmap
        fid, cb = struct.unpack_from("<II", r, 0)
        return fid == frame_id and len(r) == 8 + cb and dec(r[8:], len(bitmap)) == bitmap
    return None


def pin_rank_cpus(local_rank, local_world):
    """N ranks share one host: each keeps to its own share of the CPUs the job may really use (affinity mask capped by the cgroup's
    quota, pyrecode_amd.misc.effective_cpus) - the first `usable // N` CPUs of the rank's N-th of the visible list, so ranks of the
    two halves of the node stay on their socket.  Must run BEFORE the first GPU call (the runtimes' helper threads inherit the mask).
    Returns the CPUs, or None when nothing was changed."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        from pyrecode_amd.misc import effective_cpus
        usable, _ = effective_cpus()
        cpus = sorted(os.sched_getaffinity(0))
        chunk = len(cpus) // local_world
        if chunk < 1:
            return None
        per = max(1, min(chunk, usable // local_world))
        mine = cpus[local_rank * chunk:local_rank * chunk + per]
        os.sched_setaffinity(0, mine)
        return mine
    except (OSError, ValueError):
        return None


def pattern_floor(N, B, k_ms):
    """What the memory system delivers for the reduce kernel's ACCESS PATTERN, from the committed probe (tools/wr_probe.hip,
    profiles/r02_reduce_stores.md: same grid, XCD mapping, frames per wave and nontemporal 16-byte loads as k_reduce_tiles, no compute):
    reads alone, and reads plus the three 128-byte lines the product writes per tile and frame.  Measured at 4096x4096 x 64 frames;
    other geometries are scaled by their bytes (and say so)."""
    path = os.path.join(REPO, "profiles", "pattern_floor.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    ref_bytes = t["bytes_read"]
    scale = (B * N * 2) / ref_bytes
    ro, rw = t["reads_only_ms"] * scale, t["reads_plus_3_lines_ms"] * scale
    return {"reads_only": round(ro, 4), "reads_plus_3_lines": round(rw, 4), "scaled_by_bytes": abs(scale - 1) > 1e-9, "source": t["source"],
            "kernel_over_floor": (round(k_ms / rw, 3) if rw > 0 else None)}


def run_rank(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    pinned = None
    if world > 1:        # before torch / HIP start their threads
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        if not a.no_pin:
            pinned = pin_rank_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    import torch
    if world > 1:
        torch.set_num_threads(1)
    import torch.distributed as dist
    from pyrecode_amd import _lib as hip
    from pyrecode_amd.parallel import ShardedStepLoop

    rank = int(os.environ.get("RANK", "0"))
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # launched by torch.distributed.run (any world size)
    if world != a.gpus:
        raise SystemExit("rank %d: --gpus %d but WORLD_SIZE=%d" % (rank, a.gpus, world))
    visible = torch.cuda.device_count()
    if a.shared_gpu:      # rehearsal on a one-GPU box: every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one GPU)
        local = 0
    if local >= visible:
        raise SystemExit("rank %d: --gpus %d needs one GPU per rank, but this node shows %d visible GPU(s) (local rank %d has none; "
                         "RCCL does not take two ranks on one device - rehearse with --shared-gpu --dist-backend gloo)" % (rank, a.gpus, visible, local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = a.dist_backend
    # one line per rank on stderr, so that a multi-GPU log explains itself: who runs where, on which CPUs, over what
    try:
        cpus = sorted(os.sched_getaffinity(0))
        cpu_txt = "%d CPUs [%s]%s" % (len(cpus), ",".join(str(c) for c in cpus[:8]) + (",..." if len(cpus) > 8 else ""), " (pinned)" if pinned else "")
    except (AttributeError, OSError):
        cpu_txt = "CPU set unknown"
    print("bench.py start: rank %d of world %d (local rank %d), visible GPUs %d, device cuda:%d (%s), launch %s, backend %s, gather every %s, %s"
          % (rank, world, local, visible, local, torch.cuda.get_device_name(local),
             "torch.distributed.run" if use_dist else "single process", (backend if use_dist else "none"),
             ("region" if a.gather_every == 0 else "%d step(s)" % a.gather_every), cpu_txt), file=sys.stderr, flush=True)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    L = hip.lib()
    N = a.ny * a.nx
    B, S = a.batch, max(a.stack, a.batch)
    S -= S % B
    # synthetic stack generated on the device; rank-distinct seed
    seed = 20261003 + rank
    dark = torch.empty(N, dtype=torch.int16, device=dev)
    stack = torch.empty((S, N), dtype=torch.int16, device=dev)
    hip.check(L.rc_synth_dark(local, seed, N, dark.data_ptr()))
    for lo in range(0, S, 64):
        n = min(64, S - lo)
        if a.clustered:
            hip.check(L.rc_synth_frames_clustered(local, seed, lo, n, a.nx, a.ny, a.sparsity_ppm, dark.data_ptr(), stack[lo].data_ptr()))
        else:
            hip.check(L.rc_synth_frames(local, seed, lo, n, N, a.sparsity_ppm, dark.data_ptr(), stack[lo].data_ptr()))

    src_dtype = np.uint16
    if a.source_bytes == 1:
        # uint8 sources: the same events on an 8-bit scale - dark 5..7, an event's residual 1..200, background below the dark level
        src_dtype = np.uint8
        dark8 = (dark >> 4).to(torch.uint8)
        stack8 = torch.empty((S, N), dtype=torch.uint8, device=dev)
        for lo in range(0, S, 16):
            f16 = stack[lo:lo + 16]
            amp = f16 - dark[None]
            stack8[lo:lo + 16] = torch.where(amp > 0, dark8[None].to(torch.int16) + 1 + amp % 200, f16 >> 5).to(torch.uint8)
        del stack, dark, f16, amp
        torch.cuda.empty_cache()
        stack, dark = stack8, dark8
    if a.source_bytes == 4:
        # uint32 sources: the same events on a 20-bit (or wider) scale - the dark level and the residuals times 64
        src_dtype = np.uint32
        stack32 = torch.empty((S, N), dtype=torch.int32, device=dev)
        for lo in range(0, S, 16):
            stack32[lo:lo + 16] = stack[lo:lo + 16].to(torch.int32) * 64
        dark32 = dark.to(torch.int32) * 64
        del stack, dark
        torch.cuda.empty_cache()
        stack, dark = stack32, dark32
    op_mode = 1
    ctx = hip.ReduceContext(a.nx, a.ny, a.depth, a.level, op_mode, a.scheme, a.clevel, local, max_batch=B, src_dtype=src_dtype, device_zlib=a.device_zlib)
    ctx.set_dark(dark.data_ptr(), 0)  # eps = 0 -> thr = dark
    ctx.keep_binary_maps(False)       # no validation frames in this workload: records only (recode_writer.py:402-415)
    out_cap = int(L.rc_out_capacity(ctx.handle, B))   # the worst case the library itself states: B raw frames (a record may not exceed its frame, recode_writer.py:565-566)
    # two sets of output buffers, alternating: a pipelined ctx may run step i + 1's second stage while step i's is still writing its records
    # (include/recode_hip.h: "the caller must not reuse a batch's output buffers before" it has ordered itself behind the batch)
    outs = [torch.empty(out_cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    recs = [torch.empty(B + 1, dtype=torch.int64, device=dev) for _ in range(2)]
    nb = S // B
    # rank r's frames of step i carry the ids of its contiguous block of the job's frames (recode_writer.py:320-322,385)
    batch_ptr = [stack[j * B].data_ptr() for j in range(nb)]      # (no tensor indexing in the step loop: N loops share the host's cores)
    collective = use_dist and not a.no_collective
    loop = ShardedStepLoop(ctx, B, lambda i: (batch_ptr[i % nb], rank * S + (i % nb) * B), outs, recs, dev, collective=collective,
                           gather_every=a.gather_every, region_steps=max(a.steps, a.warmup, 1), fence_barrier=use_dist)
    stream = loop.stream
    if not os.environ.get("RC_BENCH_OWN_STREAM"):     # (experiments with a CU-masked stream of the library's own: RC_RSTREAM_EXCL)
        ctx.set_stream(stream.cuda_stream)
    ctx.set_pipelined(not a.no_pipeline)  # batch i+1's reduce kernel may overlap batch i's scans / layout / assembly
    step, fence = loop.step, loop.fence

    # The timed region is EXACTLY a.steps steps between two fences (barrier + synchronize on both sides).  It is repeated
    # (>= 3 times, until --min-seconds of timed work): at 0.5 ms per step a single pass of the driver's --steps 20 lasts 10 ms,
    # too short for the clocks to settle or for the driver's utilisation sampler to see the GPU.  The MEDIAN repeat is reported.
    with torch.cuda.stream(stream):
        for i in range(a.warmup):
            step(i)
        ctx.sync()
        ctx.set_profiling(True, every=a.kernel_events_every)
        times, k_ms_list, it, issue = [], [], a.warmup, []
        more = True
        while more:
            fence()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(it + i)
            t_issued = time.perf_counter()
            fence()
            dt = time.perf_counter() - t0
            issue.append((t_issued - t0) / a.steps)
            it += a.steps
            ctx.sync()  # also raises if the device flagged a batch
            sums, nbatches = ctx.profile()
            ctx.set_profiling(True, every=a.kernel_events_every)   # clears the sums for the next repeat
            assert nbatches == -(-a.steps // a.kernel_events_every)   # every k-th launch of the region carries events, the first included
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            times.append(float(t.item()))   # the same list on every rank, so every rank leaves the loop together
            k_ms_list.append((sums[0] / nbatches, sums[4] / nbatches, [v / nbatches for v in sums]))
            more = len(times) < 3 or (sum(times) < a.min_seconds and len(times) < 1000)
        # host side of a step by itself: a short burst into an EMPTY queue (nothing to wait for, no back-pressure), no sync inside
        ctx.set_profiling(False)
        burst, enq = min(16, max(a.steps, 1)), []
        for _ in range(5):
            fence()
            t0 = time.perf_counter()
            for i in range(burst):
                step(it + i)
            enq.append((time.perf_counter() - t0) / burst)
            it += burst
        fence()
    host_enqueue_us = sorted(enq)[len(enq) // 2] * 1e6
    order = sorted(range(len(times)), key=lambda i: times[i])
    mid = order[len(order) // 2]
    dt_max = times[mid]
    sums = [v * a.steps for v in k_ms_list[mid][2]]
    nbatches = a.steps
    issue_us = issue[mid] * 1e6
    last_step = it - 1
    frames_total = world * B * a.steps
    fps = frames_total / dt_max
    rec_h = loop.rec.cpu().numpy()
    assert rec_h[0] == 0 and rec_h[-1] > 0
    gather_verified = loop.verify_gather()   # collective: every rank's block of the gathered table, on every rank
    gathers_in_region = loop.gathers_issued

    thr_h = dark.cpu().numpy().view(src_dtype).astype(np.uint16 if a.source_bytes < 4 else np.uint32)
    corrupt = os.environ.get("RC_BENCH_CORRUPT_RECORD")          # test switch: one byte of a record flipped before the check

    def verify_batch(j, z):
        """Outside the timing: record z of the records `out` holds now - batch j of the stack - against the oracle."""
        out, rec = loop.out, loop.rec         # (the buffers the most recent step wrote)
        rec_now = rec.cpu().numpy()
        if corrupt:
            mid_byte = (int(rec_now[z]) + int(rec_now[z + 1])) // 2
            out[mid_byte] ^= 0x5A
        frame = stack[j * B + z].cpu().numpy().view(src_dtype)
        frame = frame.astype(np.uint16) if a.source_bytes < 4 else frame   # (uint8 sources: widened - same values - for the uint16 oracle)
        r = out[int(rec_now[z]):int(rec_now[z + 1])].cpu().numpy().tobytes()
        try:
            return bool(verify_record(a, r, frame, thr_h, rank * S + j * B + z))
        except Exception as e:   # a record the stock decoder rejects is a failed check, not a crashed bench (and every rank goes on
            print("rank %d: verification of batch %d record %d raised: %r" % (rank, j, z, e), file=sys.stderr)   # through the same steps)
            return False

    # One record of EVERY distinct batch of the stack: the last timed step's records as they stand, then each other batch run once more
    # through the same loop (same kernels, same buffers; every rank takes the same steps, so the collectives stay matched).
    checked, verified = [], True
    try:
        j_last = last_step % nb
        zs = [(B // 2 + 7 * k) % max(B - 1, 1) for k in range(nb)]   # one record inside the batch ...

        def check_two(j, z):   # ... and the batch's LAST record (behind every other record's bytes: a wrong size anywhere in the batch moves it)
            ok_all = True
            for zz in sorted({z, B - 1}):
                ok1 = verify_batch(j, zz)
                checked.append({"batch": j, "record": zz, "ok": ok1})
                ok_all = ok_all and ok1
            return ok_all
        verified = check_two(j_last, zs[0]) and verified
        with torch.cuda.stream(stream):
            for k in range(1, nb):
                j = (j_last + k) % nb
                i_extra = it + ((j - it) % nb)                   # a step index that maps to batch j
                fence()
                step(i_extra)
                fence()
                ctx.sync()
                verified = check_two(j, zs[k]) and verified
    except Exception as e:
        verified = False
        print("rank %d: verification raised: %r" % (rank, e), file=sys.stderr)
        if use_dist:
            raise            # (a rank that leaves the steps the others take would leave them waiting at a fence)
    if use_dist:   # every rank checks one of ITS records; the line reports the conjunction
        v = torch.tensor([1 if verified else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(int(v.item()) == 1)

    if rank == 0:
        k_ms = sums[0] / max(nbatches, 1)
        frame_bytes = N * a.source_bytes        # algorithmic bytes: one read of the frame in its source dtype
        achieved = B * frame_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_note = None, None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        key = "%dx%d_b%d_ppm%d_d%d_s%d%s%s%s%s" % (a.ny, a.nx, B, a.sparsity_ppm, a.depth, a.scheme, "_clustered" if a.clustered else "", {1: "_u8", 2: "", 4: "_u32"}[a.source_bytes],
                                              "" if a.level == 1 else "_l%d" % a.level, "_devzlib" if a.device_zlib and a.scheme == 0 else "")   # (the reduce kernel of levels 2 / 3 writes other things)
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(key, {}).get("reduce_kernel_hbm_bytes_per_launch")
        if traffic is None:
            traffic_note = "no PMC pass committed for configuration %s (profiles/traffic.json)" % key
        else:
            traffic_note = "PMC (2*FETCH_SIZE + WRITE_SIZE) of a separate rocprofv3 --pmc run of this configuration, profiles/traffic.json"
        result = {
            "metric": "frames/sec + GB/s in, 4096x4096 uint16 @1% sparsity, 1/2/4/8 GPU",
            "value": round(fps, 1), "unit": "frames/s", "gb_per_s_in": round(fps * frame_bytes / 1e9, 1),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt_max / a.steps * 1e3, 4),
            "repeats": len(times), "timed_seconds_total": round(sum(times), 3),
            "ms_per_step_all_repeats": {"min": round(min(times) / a.steps * 1e3, 4), "median": round(dt_max / a.steps * 1e3, 4),
                                        "max": round(max(times) / a.steps * 1e3, 4)},
            "verified": verified, "records_checked": checked, "gather_verified": gather_verified,
            # host side of one step (Python loop + ctypes + launches + the collective's enqueue): a burst of steps issued into an empty
            # queue without any sync; `issue_us_per_step_in_timed_region` is the same loop inside the timed region (includes waiting
            # for room in the device queue when the host runs ahead)
            "host_enqueue_us_per_step": round(host_enqueue_us, 1), "issue_us_per_step_in_timed_region": round(issue_us, 1),
            "host_enqueue_frac_of_step": round(host_enqueue_us / (dt_max / a.steps * 1e6), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {1: "u8", 2: "u16", 4: "u32"}[a.source_bytes], "data": "synthetic",
            "config": {
                "workload": "%s%dx%d %s, %s, L%d + %s, source_bit_depth %d, batch %d frames/GPU/step, %d-frame stack/GPU in HBM" % (
                    ("BASELINE configs[%d]: " % (a.config - 1)) if a.config else "", a.ny, a.nx, {1: "uint8", 2: "uint16", 4: "uint32"}[a.source_bytes],
                    ("detector-like clusters of 1..6 pixels, %d seeds per million pixels" % a.sparsity_ppm) if a.clustered else "%.2f%% sparsity" % (a.sparsity_ppm / 1e4), a.level,
                    {2: "LZ4 frame", 1: "zstd frame (%s encoder)" % ("modelled" if a.clevel else "fast"), 8: "blosc-lz4 chunk",
                     0: ("zlib streams from the device DEFLATE encoder (compression_scheme 0)" if a.device_zlib else
                         "reduce-only pieces (the host library compresses them, as the reference does)")}.get(a.scheme, str(a.scheme)),
                    a.depth, B, S),
                "parallelism": "dp%d (contiguous frame blocks per rank; %s)" % (world, (
                    "no collective: the ranks meet only at the fences (host-side rehearsal)" if use_dist and not collective else
                    ("%s all-gather of the metadata rows %s, on a side stream under the following steps" % (
                        {"nccl": "RCCL"}.get(backend, backend), "once per timed region" if a.gather_every == 0 else
                        ("every step" if a.gather_every == 1 else "every %d steps" % a.gather_every))) if use_dist else "single process: no collective")),
                "gather_every": a.gather_every, "gathers_issued_in_all": gathers_in_region, "cpus_pinned": (len(pinned) if pinned else None),
                "record_bytes_per_frame": round(float(rec_h[-1]) / B, 1),
                "launch": ("bench.py started its own ranks (child torch.distributed.run)" if os.environ.get("RC_BENCH_SELF_LAUNCHED")
                           else ("torch.distributed.run" if use_dist else "single process")),
                "collective_backend": (dist.get_backend() if use_dist else None), "collective_ranks": (dist.get_world_size() if use_dist else 1),
                "shared_gpu_rehearsal": bool(a.shared_gpu),
            },
            "rccl_ranks": (dist.get_world_size() if use_dist and backend == "nccl" else (1 if not use_dist else 0)),
            "roofline": {"bound": "hbm", "kernel": "k_reduce_tiles32" if a.source_bytes == 4 else "k_reduce_tiles", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_ratio": (round(traffic / (B * frame_bytes), 4) if traffic else None),
                         "kernel_ms": round(k_ms, 4), "kernel_events_every": a.kernel_events_every, "algorithmic_bytes_per_launch": B * frame_bytes,
                         "pattern_floor_ms": pattern_floor(N, B, k_ms) if a.source_bytes == 2 else None,
                         "whole_path_frac": round(fps / world * frame_bytes / 1e9 / HBM_PEAK_GBS, 4)},
            # only the events the roofline needs are recorded in the timed region (each costs stream time); the full
            # per-stage split is available with RC_PROFILE_ALL_STAGES=1
            "stage_ms_per_step": ({"reduce": round(sums[0] / nbatches, 4), "bitmap_codec_kernel": round(sums[1] / nbatches, 4),
                                   "scan": round(sums[2] / nbatches, 4), "layout_assemble": round(sums[3] / nbatches, 4),
                                   "total": round(sums[4] / nbatches, 4)} if os.environ.get("RC_PROFILE_ALL_STAGES") else
                                  {"reduce": round(sums[0] / nbatches, 4), "everything_else": round((sums[4] - sums[0]) / nbatches, 4),
                                   "total": round(sums[4] / nbatches, 4)}),
        }
        if world == 1 and not a.no_cpu_baseline:
            ns = min(32, S)
            frames_h = stack[:ns].cpu().numpy().view(np.uint16)
            thr_h = dark.cpu().numpy().view(np.uint16)
            result["cpu_baseline"] = cpu_baseline(frames_h, thr_h, a.depth, a.scheme, zlib_level=(a.clevel if a.scheme == 0 and a.device_zlib else None))
        else:
            result["cpu_baseline"] = None
        if world == 1 and not a.no_ingest and a.level in (1, 3):
            try:
                result["ingest_inclusive"], host_frames = ingest_inclusive(stack, dark, a, nframes=a.ingest_frames)
                v, _ = ingest_inclusive(stack, dark, a, nframes=a.ingest_frames, validation_frame_gap=10, data=host_frames)   # validation frames ride the same stream (recode_writer.py:402-415)
                del host_frames
                result["ingest_inclusive"]["with_validation_frame_gap_10"] = {
                    "frames_per_s": v["frames_per_s"], "gb_per_s_in": v["gb_per_s_in"], "frames_per_s_min_max": v["frames_per_s_min_max"],
                    "ratio_to_plain": round(v["frames_per_s"] / max(result["ingest_inclusive"]["frames_per_s"], 1e-9), 3)}
            except Exception as e:   # an extra: never lets the contract line fail
                result["ingest_inclusive"] = {"error": repr(e)}
        if world == 1 and not a.no_ingest and a.level in (1, 3) and a.scheme in (1, 2):
            # extra, not `value`: the READER's batched path on the same configuration (stored frames in host memory -> device decode of both
            # streams -> triplets in device memory; bench.py --read prints this as a line of its own)
            try:
                import copy
                ra = copy.copy(a)
                ra.steps, ra.warmup, ra.min_seconds = 20, 3, 0.5
                ctx.close()
                del stack
                torch.cuda.empty_cache()
                rl = bench_read(ra, emit=False)
                result["reader"] = {"frames_per_s_streaming": rl["value"], "frames_per_s_one_call_at_a_time": rl["one_call_at_a_time"]["frames_per_s"],
                                    "verified": rl["verified"], "what": rl["metric"]}
            except Exception as e:
                result["reader"] = {"error": repr(e)}
        print(json.dumps(result), flush=True)
    ok = (gather_verified is not False)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if not ok:
        raise SystemExit("rank %d: the gathered metadata table does not hold every rank's rows" % rank)
    if not verified:     # a fast wrong answer is not a result: the line above says verified: false, the exit code says it too
        raise SystemExit("rank %d: a record failed the check against the oracle (verified: false)" % rank)


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.read:
        return bench_read(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_children(a, argv))
    run_rank(a)


if __name__ == "__main__":
    main()
