"""`c_recode.Reader` look-alike backed by the HIP library (seam 3 of SURVEY.md §8b).

The reference's only native module is the CPython extension `c_recode` (pyrecode/pyrecode.cpp:143-150, loops in
pyrecode/c_extensions/reader.h).  This shim keeps its four method names and argument order so ReCoDeReader /
ReCoDeWriter code paths are unchanged, and routes them to rc_unpack_frame_sparse / rc_bit_pack / rc_bit_unpack.
Differences are the reference's defects: arguments are validated, failures raise instead of
returning a str, the packer zeroes its output, the unpacker terminates."""
import ctypes as C

import numpy as np

from . import _lib


def _addr(buf, writable=False):
    """(address, nbytes, keepalive) of any bytes-like object."""
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data, buf.nbytes, buf
    mv = memoryview(buf)
    if writable and mv.readonly:
        # the reference hands in memoryview(bytes(...)) as an output buffer (recode_reader.py:115) and writes through it
        arr = np.frombuffer(mv, dtype=np.uint8)
        return arr.ctypes.data, arr.nbytes, arr
    arr = np.frombuffer(mv, dtype=np.uint8)
    return arr.ctypes.data, arr.nbytes, arr


class Reader:

    def __init__(self):
        self.ny = self.nx = self.bit_depth = 0

    def create_buffers(self, ny, nx, bit_depth):
        """pyrecode.cpp:60-76: remembers the frame shape and the field width.  Returns 1."""
        self.ny, self.nx, self.bit_depth = int(ny), int(nx), int(bit_depth)
        return 1

    def get_frame_sparse(self, reduction_level, bitmap, pixvals, out):
        """pyrecode.cpp:95-119 -> reader.h:10-68: (row, col, val) uint64 triplets into `out`; returns nnz."""
        pb, nb, k1 = _addr(bitmap)
        need = (self.nx * self.ny + 7) // 8
        if nb < need:
            raise ValueError("binary map shorter than ceil(nx*ny/8) bytes")
        if pixvals is None or reduction_level != 1:
            pp, npx, k2 = None, 0, None
        else:
            pp, npx, k2 = _addr(pixvals)
            if npx == 0:
                pp = None
        po, no, k3 = _addr(out, writable=True)
        n = _lib.lib().rc_unpack_frame_sparse(self.nx, self.ny, self.bit_depth, pb, pp, npx, po, no // 24, int(reduction_level))
        return _lib.check(n, "get_frame_sparse")

    def count(self, bitmap):
        """Set pixels of a packed binary map (counting pass of rc_unpack_frame_sparse; sizes the triplet buffer)."""
        pb, nb, k1 = _addr(bitmap)
        if nb < (self.nx * self.ny + 7) // 8:
            raise ValueError("binary map shorter than ceil(nx*ny/8) bytes")
        return _lib.check(_lib.lib().rc_unpack_frame_sparse(self.nx, self.ny, self.bit_depth, pb, None, 0, None, 0, 3), "count")

    def bit_pack_pixel_intensities(self, sz_packed, n_pixels, bit_depth, pixvals, out):
        """pyrecode.cpp:121-141 -> reader.h:105-140.  Returns the elapsed milliseconds like the reference (here 0.0:
        stage times come from HIP events, ReCoDeWriter.run())."""
        pv, nv, k1 = _addr(pixvals)
        po, no, k2 = _addr(out, writable=True)
        if nv < 2 * n_pixels or no < sz_packed:
            raise ValueError("buffer too small")
        _lib.check(_lib.lib().rc_bit_pack(pv, int(n_pixels), int(bit_depth), po, int(sz_packed)), "bit_pack_pixel_intensities")
        return 0.0

    def bit_unpack_pixel_intensities(self, n_values, packed, out):
        """pyrecode.cpp:78-93 -> reader.h:74-99 (intended semantics): n_values fields of self.bit_depth bits -> uint64."""
        pp, npk, k1 = _addr(packed)
        po, no, k2 = _addr(out, writable=True)
        if no < 8 * int(n_values):
            raise ValueError("buffer too small")
        _lib.check(_lib.lib().rc_bit_unpack(pp, npk, int(n_values), self.bit_depth, po), "bit_unpack_pixel_intensities")
        return int(n_values)
