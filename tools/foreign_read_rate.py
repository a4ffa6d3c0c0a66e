"""Development: rate of the batched reader on a file whose streams a STOCK encoder wrote (what the reference's writer produces with
zstandard / lz4.frame): host decode on the thread pool + one device expand per batch, against the frame-at-a-time API.
usage: foreign_read_rate.py [scheme 1|2] [nframes] [batch]"""
import ctypes as C, ctypes.util, os, struct, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyrecode_amd import synth
from pyrecode_amd.params import InputParams
from pyrecode_amd.recode_reader import ReCoDeReader, merge_parts
from pyrecode_amd.recode_writer import ReCoDeWriter

scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 32
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ny = nx = 4096
N = ny * nx
if scheme == 1:
    z = C.CDLL(ctypes.util.find_library("zstd"))
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    def enc(b):
        dst = C.create_string_buffer(len(b) + len(b) // 8 + 1024)
        n = z.ZSTD_compress(dst, len(dst), b, len(b), 1)
        return dst.raw[:n]
else:
    lz = C.CDLL(ctypes.util.find_library("lz4"))
    lz.LZ4F_compressFrameBound.restype = C.c_size_t
    lz.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
    lz.LZ4F_compressFrame.restype = C.c_size_t
    lz.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    def enc(b):
        dst = C.create_string_buffer(lz.LZ4F_compressFrameBound(len(b), None) + 64)
        n = lz.LZ4F_compressFrame(dst, len(dst), b, len(b), None)
        return dst.raw[:n]
dark = synth.dark_frame(3, N)
frames = synth.frames(3, 0, min(nz, 8), N, 10000, dark)
ip = InputParams()
ip._param_map.update(dict(reduction_level=1, rc_operation_mode=1, calibration_threshold_epsilon=0, target_bit_depth=16, source_bit_depth=16,
                          num_cols=nx, num_rows=ny, num_frames=1, frame_offset=0, num_calibration_frames=1, calibration_frame_offset=0,
                          keep_part_files=1, num_threads=1, l2_statistics=0, l4_centroiding=0, compression_scheme=scheme, compression_level=1,
                          source_file_type=0, source_header_length=0, keep_calibration_data=0, calibration_file_type=0, source_data_type=0,
                          target_data_type=0))
tmp = tempfile.mkdtemp(dir="/dev/shm")
w = ReCoDeWriter("own.bin", dark_data=dark.reshape(ny, nx), output_directory=tmp, input_params=ip, mode="batch", node_id=0)
w.start(); w.run(frames[:1].reshape(1, ny, nx)); w.close()
merge_parts(tmp, "own.rc1", 1)
hdr = bytearray(open(os.path.join(tmp, "own.rc1"), "rb").read()[:512])
hdr[23:27] = struct.pack("<I", nz)
md, blobs = [], []
for k in range(frames.shape[0]):
    f = frames[k]
    mask = f > dark
    bitmap = np.packbits(mask, bitorder="little").tobytes()
    packed = (f[mask] - dark[mask]).astype("<u2").tobytes()
    cb, cp = enc(bitmap), enc(packed)
    md.append(struct.pack("<III", len(cb), len(cp), len(packed)))
    blobs.append(cb + cp)
path = os.path.join(tmp, "foreign.rc1")
with open(path, "wb") as fo:
    fo.write(bytes(hdr))
    for k in range(nz):
        fo.write(md[k % len(md)])
    for k in range(nz):
        fo.write(blobs[k % len(blobs)])
rd = ReCoDeReader(path, is_intermediate=False)
rd.open(print_header=False)
rd.get_frames_triplets(0, min(batch, nz))
t0 = time.perf_counter()
for a in range(0, nz, batch):
    rd.get_frames_triplets(a, min(batch, nz - a))
dt = time.perf_counter() - t0
print("[foreign] scheme %d, %d frames 4096x4096 1 %%, batches of %d: %.0f frames/s (%s)" % (scheme, nz, batch, nz / dt, rd.last_batch_path))
import pyrecode_amd.recode_reader as RR
acc = dict(decode=0.0)
_orig = rd._host_decode_batch
def timed(*a, **k):
    t = time.perf_counter(); r = _orig(*a, **k); acc["decode"] += time.perf_counter() - t; return r
rd._host_decode_batch = timed
from pyrecode_amd import _lib as _L
LL = _L.lib()
_sub, _wait = LL.rc_expand_frames_submit, LL.rc_expand_frames_wait
def t_sub(*a):
    acc["t_sub"] = time.perf_counter(); return _sub(*a)
def t_wait(*a):
    r = _wait(*a); acc["device"] = acc.get("device", 0.0) + time.perf_counter() - acc["t_sub"]; return r
LL.rc_expand_frames_submit, LL.rc_expand_frames_wait = t_sub, t_wait
t0 = time.perf_counter()
nseen = 0
for a, pre, tr in rd.iter_frames_triplets(0, nz, batch=batch):
    nseen += len(pre) - 1
dt = time.perf_counter() - t0
print("[foreign] streaming iterator (decode one batch ahead of the device): %.0f frames/s; host decode alone %.0f frames/s, device copy-in + expand + copy-out alone %.0f frames/s (%s)"
      % (nseen / dt, nseen / max(acc["decode"], 1e-9), nseen / max(acc.get("device", 0), 1e-9), rd.last_batch_path))
LL.rc_expand_frames_submit, LL.rc_expand_frames_wait = _sub, _wait
rd._host_decode_batch = _orig
for rep in range(2):
    t0 = time.perf_counter()
    nseen = 0
    for a, pre, (rows, cols, vals) in rd.iter_frames_coo(0, nz, batch=batch):
        nseen += len(pre) - 1
    dt = time.perf_counter() - t0
print("[foreign] streaming iterator, COO layout (10 bytes a set pixel to the host): %.0f frames/s" % (nseen / dt))
t0 = time.perf_counter()
for zf in range(nz):
    rd.get_frame(zf)
dt = time.perf_counter() - t0
print("[foreign] frame-at-a-time get_frame over the whole file (reads ahead once sequential): %.0f frames/s" % (nz / dt))
rd._ra_off = True
rd._drop_readahead()
t0 = time.perf_counter()
for zf in range(min(nz, 8)):
    rd.get_frame(zf)
dt = time.perf_counter() - t0
print("[foreign] frame-at-a-time get_frame, one frame per call: %.0f frames/s" % (min(nz, 8) / dt))
rd.close()
import shutil; shutil.rmtree(tmp, ignore_errors=True)
