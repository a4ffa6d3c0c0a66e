"""Source-file readers for the writer's file mode: MRC / MRCS stacks and Norpix StreamPix .seq files.

Interface = the reference's pyrecode/em_reader.py (emfile :11-33, EMReaderBase :36-184, MRCReader :187-240, SEQReader
:243-302): `emfile(path, file_type)` context manager, `.shape` (nz, ny, nx), `.dtype`, `.header` dict, `reader[z]` ->
array [1, ny, nx], `reader[z0:z1]` -> [n, ny, nx], `.serialize_header(fp)` (1024 bytes), `.close()`.

The reference delegates the parsing to two third-party packages, `mrcfile` and `pims`, neither pinned in its
requirements nor present in the build image (ImportError, not a refusal).  The formats are therefore read directly
from their published layouts with numpy memory maps - no copy of the stack is made until frames are sliced out:

MRC2014 (CCP-EM): 1024-byte header of little- or big-endian words (machine stamp at byte 212): nx, ny, nz, mode at
  words 0-3 (mode 0 int8, 1 int16, 2 float32, 6 uint16, 12 float16), extended-header length `nsymbt` at word 23,
  "MAP " at byte 208; data follow the extended header, x fastest.  Parity: unpinned against mrcfile (absent); the
  tests write files with an independent packer and compare.
Norpix SEQ (StreamPix): little-endian header: magic 0xFEED at 0, name "Norpix seq" (UTF-16) at 4, version int32 at
  28, header size int32 at 32 (1024), image info at 548: width, height, bit depth, real bit depth, image size in
  bytes, image format (all uint32), then allocated frames at 572, origin at 576, true image size at 580 (the stride
  between frames: image bytes + 8-byte timestamp, padded), frame rate double at 584.  Frames start at byte 8192 for
  version >= 5 files, else at 1024.  Parity: unpinned against pims (absent).
"""
import os
import struct

import numpy as np

from .misc import rc_cfg as rc

DEFAULT_BUFFER_SIZE = 8 * 1024  # kept for signature compatibility (reference :8)


def emfile(file, file_type=None, mode="r", buffering=-1):
    """Reference em_reader.py:11-33.  file_type: rc.FILE_TYPE_MRC / rc.FILE_TYPE_SEQ."""
    if mode != "r":
        raise NotImplementedError("emfile supports only 'r' mode.")
    if file_type == rc.FILE_TYPE_MRC:
        return MRCReader(file)
    if file_type == rc.FILE_TYPE_SEQ:
        return SEQReader(file)
    if file_type == rc.FILE_TYPE_BINARY:
        raise NotImplementedError
    raise ValueError("Source type: %s is not supported." % (file_type,))


class EMReaderBase:
    """Common slicing / iteration protocol (reference :36-184)."""

    def __init__(self, file, source_type='', fast_random_access=False, buffer_size=DEFAULT_BUFFER_SIZE):
        self._source_filename = file
        self._source_type = source_type
        self._open()
        self._header = self._load_header()
        self._shape = self._get_shape()
        self._dtype = self._get_dtype()
        self.buffer_size = buffer_size
        self._fast_random_access = fast_random_access
        self._current_z = 0

    source_type = property(lambda self: self._source_type)
    shape = property(lambda self: self._shape)
    header = property(lambda self: self._header)
    dtype = property(lambda self: self._dtype)
    fast_random_access = property(lambda self: self._fast_random_access)

    def __iter__(self):
        return self

    def __next__(self):
        if self._current_z >= self.shape[0]:
            raise StopIteration
        self._current_z += 1
        return self._get_frame(self._current_z - 1)

    def __getitem__(self, key):
        full_y, full_x = slice(0, self._shape[1]), slice(0, self._shape[2])
        if isinstance(key, tuple):
            if len(key) == 3:
                return self._get_sub_volume(key[0], key[1], key[2])
            if len(key) == 2:
                return self._get_sub_volume(key[0], key[1], full_x)
            if len(key) == 1:
                return self._get_sub_volume(key[0], full_y, full_x)
            raise TypeError
        if isinstance(key, slice):
            return self._get_sub_volume(key, full_y, full_x)
        if isinstance(key, (int, np.integer)):
            return self._get_frame(int(key))
        raise TypeError

    def __enter__(self):
        return self

    def __exit__(self, exc_type, value, traceback):
        self.close()

    def print_header(self):
        for field in self._header:
            print(field + ":\t" + str(self._header[field]))

    def _get_frame(self, z_index):
        if not 0 <= z_index < self.get_true_shape()[0]:
            raise IndexError("frame %d outside the %d frames in the file" % (z_index, self.get_true_shape()[0]))
        return np.asarray(self._stack[z_index])[np.newaxis, :, :]

    def _get_sub_volume(self, slice_z, slice_y, slice_x):
        # like the reference (which indexes the package's array), a z range that runs past the data raises IndexError
        # so that the writer falls back to frame-by-frame loading (recode_writer.py:333-348)
        n = self.get_true_shape()[0]
        if isinstance(slice_z, slice) and slice_z.stop is not None and slice_z.stop > n:
            raise IndexError("frames %s requested, %d in the file" % (slice_z, n))
        return np.asarray(self._stack[slice_z, slice_y, slice_x])

    def get_true_shape(self):
        return self._stack.shape

    def close(self):
        self._stack = None


_MRC_MODES = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16, 12: np.float16}


class MRCReader(EMReaderBase):
    def __init__(self, file):
        EMReaderBase.__init__(self, file, 'mrc', True)

    def _open(self):
        with open(self._source_filename, 'rb') as fp:
            self._raw_header = fp.read(1024)
        if len(self._raw_header) < 1024:
            raise ValueError("%s: shorter than an MRC header" % self._source_filename)
        stamp = self._raw_header[212]
        self._bo = '>' if stamp == 0x11 else '<'   # 0x44 0x44 (or 0x44 0x41) little endian, 0x11 0x11 big endian
        w = struct.unpack(self._bo + '56i', self._raw_header[:224])
        nx, ny, nz, mode = w[0], w[1], w[2], w[3]
        if mode not in _MRC_MODES or nx <= 0 or ny <= 0 or nz <= 0:
            raise ValueError("%s: unsupported MRC header (nx %d ny %d nz %d mode %d)" % (self._source_filename, nx, ny, nz, mode))
        self._words = w
        dt = np.dtype(_MRC_MODES[mode]).newbyteorder(self._bo)
        offset = 1024 + max(int(w[23]), 0)
        avail = max(0, (os.path.getsize(self._source_filename) - offset) // (nx * ny * dt.itemsize))
        nz_true = min(nz, avail)
        self._stack = np.memmap(self._source_filename, dtype=dt, mode='r', offset=offset, shape=(nz_true, ny, nx)) if nz_true else \
            np.zeros((0, ny, nx), dt)

    def _load_header(self):
        w = self._words
        names = ('nx', 'ny', 'nz', 'mode', 'nxstart', 'nystart', 'nzstart', 'mx', 'my', 'mz')
        h = dict(zip(names, w[:10]))
        h['ispg'], h['nsymbt'] = w[22], w[23]
        h['map'] = self._raw_header[208:212]
        h['machst'] = self._raw_header[212:216]
        h['dmin'], h['dmax'], h['dmean'] = struct.unpack(self._bo + '3f', self._raw_header[76:88])
        h['cella'] = struct.unpack(self._bo + '3f', self._raw_header[40:52])
        return h

    def _get_shape(self):
        return (self._header['nz'], self._header['ny'], self._header['nx'])

    def _get_dtype(self):
        return self._stack.dtype

    def serialize_header(self, fp):
        fp.write(self._raw_header)


class SEQReader(EMReaderBase):
    def __init__(self, file, buffer_size=DEFAULT_BUFFER_SIZE):
        EMReaderBase.__init__(self, file, 'seq', False, buffer_size)

    def _open(self):
        with open(self._source_filename, 'rb') as fp:
            raw = fp.read(1024)
        if len(raw) < 1024 or struct.unpack('<I', raw[:4])[0] != 0xFEED:
            raise ValueError("%s: not a Norpix sequence file" % self._source_filename)
        version, header_size = struct.unpack('<ii', raw[28:36])
        width, height, bit_depth, bit_depth_real, image_bytes, image_format = struct.unpack('<6I', raw[548:572])
        allocated, origin, true_image_size = struct.unpack('<3I', raw[572:584])
        fps, = struct.unpack('<d', raw[584:592])
        self._hdr = {'magic': 0xFEED, 'name': raw[4:28].decode('utf-16-le', 'ignore').rstrip('\x00'), 'version': version,
                     'header_size': header_size, 'description': raw[36:548].decode('utf-16-le', 'ignore').rstrip('\x00'),
                     'width': width, 'height': height, 'bit_depth': bit_depth, 'bit_depth_real': bit_depth_real,
                     'image_size_bytes': image_bytes, 'image_format': image_format, 'allocated_frames': allocated,
                     'origin': origin, 'true_image_size': true_image_size, 'suggested_frame_rate': fps}
        if bit_depth == 8:
            dt = np.dtype(np.uint8)
        elif bit_depth == 16:
            dt = np.dtype('<i2')   # the reference maps 16-bit sequences to int16 (em_reader.py:286-293)
        else:
            raise TypeError("Sequence datasets with bit-depth %d is not supported." % bit_depth)
        if width == 0 or height == 0 or image_bytes < width * height * dt.itemsize or true_image_size < image_bytes:
            raise ValueError("%s: inconsistent image geometry in the sequence header" % self._source_filename)
        first = 8192 if version >= 5 else 1024
        avail = max(0, (os.path.getsize(self._source_filename) - first + (true_image_size - image_bytes)) // true_image_size)
        n = min(allocated, avail) if allocated else avail
        if n:
            # frames are true_image_size apart; only the first width*height pixels of each stride are image data
            base = np.memmap(self._source_filename, dtype=np.uint8, mode='r', offset=first)
            self._stack = _SeqStack(base, n, height, width, dt, true_image_size)
        else:
            self._stack = _SeqStack(np.zeros(0, np.uint8), 0, height, width, dt, true_image_size)

    def _load_header(self):
        return self._hdr

    def _get_shape(self):
        return (self._header['allocated_frames'], self._header['height'], self._header['width'])

    def _get_dtype(self):
        return self._stack.dtype

    def serialize_header(self, fp):
        fp.write(bytes(1024))   # what the reference stores for sequences (em_reader.py:300-304)


class _SeqStack:
    """[n, height, width] view of a sequence file's frames (fixed stride, trailing timestamp skipped)."""

    def __init__(self, base, n, height, width, dtype, stride):
        self._base, self.shape, self.dtype, self._stride = base, (n, height, width), dtype, stride

    def _frame(self, z):
        n, h, w = self.shape
        lo = z * self._stride
        return self._base[lo:lo + h * w * self.dtype.itemsize].view(self.dtype).reshape(h, w)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            z = int(key)
            if z < 0:
                z += self.shape[0]
            if not 0 <= z < self.shape[0]:
                raise IndexError(key)
            return self._frame(z)
        if isinstance(key, tuple):
            kz, ky, kx = (key + (slice(None),) * 3)[:3]
        else:
            kz, ky, kx = key, slice(None), slice(None)
        zs = range(*kz.indices(self.shape[0])) if isinstance(kz, slice) else [int(kz)]
        out = [self._frame(z)[ky, kx] for z in zs]
        return np.stack(out) if out else np.zeros((0,) + self._frame_shape(ky, kx), self.dtype)

    def _frame_shape(self, ky, kx):
        return np.zeros(self.shape[1:], np.uint8)[ky, kx].shape
