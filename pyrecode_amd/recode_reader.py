"""ReCoDeReader / merge_parts: the reference's reader API over the HIP sparse-expand and codecs.

Same classes, method names and return shapes as reference pyrecode/recode_reader.py (ReCoDeReader :15-492,
merge_parts :495-595): get_frame / get_next_frame return {frame_id: {'metadata': {...}, 'data': scipy COO}}.  The
bitmap + packed-residual expansion runs on the GPU through the c_recode.Reader shim (rc_unpack_frame_sparse);
decompression goes through recode_compressors.de_compress (device codec or the reference's host library).
Reference defects that made files unreadable are not reproduced: L3/L4 frames can be read,
zstd frames are stream-decoded, merge_parts is a plain k-way merge by frame id.
"""
import heapq
import os

import numpy as np
from scipy.sparse import coo_matrix

from . import recode_compressors as compressors
from . import _lib
from . import c_recode
from .misc import map_dtype
from .reader_batched import BatchedAccess, _BatchOut
from .recode_header import ReCoDeHeader
from .structures import ReCoDeStructures


class ReCoDeReader(BatchedAccess):

    def __init__(self, file, is_intermediate=False):
        self._source_filename = file
        self._current_frame_index = 0
        self._c_reader = c_recode.Reader()
        self._is_intermediate = 1 if is_intermediate else 0
        self._file_size = None
        self._header = None
        self._frame_metadata = None
        self._seek_table = None
        self._rc_header = None
        self._frame_data_start_position = 0
        self._sz_frame_metadata = None
        self._n_elements_frame_metadata = None
        self._fp = None
        self._structures = None
        self._numpy_dtype = None
        self._decompressor_context = None
        # batched access (rc_expand_frames & co.): buffers, pools and bookkeeping, created on first use and released by close()
        self._pin_blob = None            # page-locked image of a batch's file bytes (get_frames_triplets)
        self._stream_bufs = None         # [blob 0, blob 1, triplets 0, triplets 1] of the streaming iterator
        self._read_pool = None           # a few threads for page-cache reads
        self._decode_pool = None         # stock decoders of the Python level (bz2, lzma; zstd / LZ4 without the shared libraries)
        self._decode_coord = None        # the thread that decodes one batch ahead (_iter_host_decoded)
        self._pin_pieces = None          # stored-pieces images of host-decoded batches: two for the iterator, one for the synchronous call
        self._host_blobs = None
        self._file_map = None            # the file, mapped, for the host decoders
        self._foreign_file = False       # the device decoders refused this file's streams once: a stock encoder wrote them
        self._no_fused_frame = False
        self.part_frame_ids = None       # part files: frame id of every record (index built on first batched access)
        self.last_batch_path = None      # 'device' | 'host-decode + device-expand' | 'per-frame'
        # read-ahead of the frame-at-a-time calls (_readahead_frame)
        self._ra = None                  # (first frame, frames, nnz prefix, triplets) of the batch fetched ahead
        self._ra_buf = None              # its page-locked triplet buffer
        self._ra_iter = None             # host-decoded files: the pipeline that decodes one batch ahead, and the frame it delivers next
        self._ra_iter_at = -1
        self._ra_last = None             # the frame asked for last
        self._ra_streak = 0              # calls in sequence so far
        self._ra_off = False             # this file gains nothing from it (or a batch failed: the per-frame path reports)
        self._user_iters = 0             # the caller's own iter_frames_* generators alive on this reader (the read-ahead stays off meanwhile)
        self.readahead_frames_served = 0

    # ---- opening -------------------------------------------------------------------------------------------
    def open(self, print_header=True):
        self._load_header(print_header)
        compressors.import_checks(self._header)
        self._fp = open(self._source_filename, "rb")
        self._fp.seek(0, 2)
        self._file_size = self._fp.tell()
        self._fp.seek(0, 0)
        self._initialize()
        self._create_read_buffers()
        self._load_seek_table()
        self._numpy_dtype = map_dtype(self._header['target_dtype'], self._header['target_bit_depth'])
        if self._header['compression_scheme'] == 1 and compressors._optional('zstandard') is not None:
            import zstandard as zstd
            self._decompressor_context = zstd.ZstdDecompressor()  # (the reference builds a ZstdCompressor here)

    def _load_header(self, print_header=True):
        self._rc_header = ReCoDeHeader()
        self._rc_header.load(self._source_filename)
        self._header = self._rc_header.as_dict()
        if print_header:
            self._rc_header.print()

    def _md_fields(self):
        return self._structures.standard_frame_metadata_structure_for(self._header['reduction_level'],
                                                                      self._header['rc_operation_mode'])

    def _initialize(self):
        h = self._header
        self._structures = ReCoDeStructures(h)
        nsm = list(self._rc_header.non_standard_metadata_sizes.values())
        self._sz_frame_metadata = self._structures.get_standard_frame_metadata_size(
            h['reduction_level'], h['rc_operation_mode']) + int(np.sum(nsm))
        self._n_elements_frame_metadata = len(nsm) + len(self._md_fields())
        self._frame_data_start_position = self._rc_header.get_frame_data_offset(self._is_intermediate,
                                                                                self._sz_frame_metadata)
        return h

    def _create_read_buffers(self):
        if 'nz' not in self._header:
            raise ValueError('Attempting to set persistent variables before reading header')
        self._c_reader.create_buffers(int(self._header["ny"]), int(self._header["nx"]), int(self._header["target_bit_depth"]))

    def _read_metadata_row(self):
        d = {}
        for field in self._md_fields():
            d[field['name']] = np.frombuffer(self._fp.read(field['bytes']), dtype=field['dtype'])[0]
        return d

    def _load_seek_table(self):
        """Merged files only: the nz-row metadata table and the cumulative data offsets (reference :127-168)."""
        if self._is_intermediate:
            return
        if 'nz' not in self._header:
            raise ValueError('Attempting to read seek table before reading header')
        h = self._header
        self._fp.seek(self._rc_header.get_frame_data_offset(True, self._sz_frame_metadata), 0)
        self._frame_metadata = [self._read_metadata_row() for _ in range(int(h['nz']))]
        sizes = np.array([self._structures.get_frame_data_size(h['reduction_level'], h['rc_operation_mode'], m)
                          for m in self._frame_metadata], dtype=np.uint64)
        self._seek_table = np.zeros((int(h['nz']), 2), dtype=np.uint64)
        self._seek_table[:, 0] = sizes
        if len(sizes) > 1:
            self._seek_table[1:, 1] = np.cumsum(sizes[:-1])

    # ---- accessors -----------------------------------------------------------------------------------------
    def get_header(self):
        return self._rc_header

    def get_source_header(self):
        return self._rc_header.source_header

    def get_true_shape(self):
        return tuple([self._header["nz"], self._header["ny"], self._header["nx"]])

    def get_shape(self):
        return tuple([self._header["nz"], self._header["ny"], self._header["nx"]])

    def get_dtype(self):
        return self._header['target_dtype']

    def get_sub_volume(self, slice_z, slice_y, slice_x):
        raise NotImplementedError

    @property
    def sz_frame_metadata(self):
        return self._sz_frame_metadata

    def __del__(self):
        try:                              # a reader dropped without close(): its read-ahead may hold a batch queued on the device
            self._close_ra_iter()
        except Exception:
            pass

    def close(self):
        self._drop_readahead()
        self._fp.close()
        if self._pin_blob is not None:
            self._pin_blob.close()
            self._pin_blob = None
        for b in self._stream_bufs or []:
            if b is not None:
                b.close()
        self._stream_bufs = None
        if self._read_pool is not None:
            self._read_pool.shutdown(wait=True)
            self._read_pool = None
        if self._decode_pool is not None:
            self._decode_pool.shutdown(wait=True)
            self._decode_pool = None
        if self._decode_coord is not None:
            self._decode_coord.shutdown(wait=True)
            self._decode_coord = None
        for b in self._pin_pieces or []:
            if b is not None:
                b.close()
        self._pin_pieces = None
        self._host_blobs = None
        if self._file_map is not None:
            try:
                self._file_map.close()
            except BufferError:                # (a caller still holds a view of a batch: the map goes with it)
                pass
            self._file_map = None

    def _read_into(self, view, pos, move_fp=True):
        """file bytes [pos, pos + len(view)) -> view (page-locked memory), in a few pieces on worker threads: a read from the page
        cache is a memcpy, and one thread moves 5-7 GB/s where the batched device reader takes 35 GB/s of compressed frames.
        move_fp=False: positional reads only - the form a helper thread uses while the caller's thread owns the file position."""
        total = view.nbytes
        nthr = 4 if total >= (8 << 20) else 1
        if nthr == 1 and move_fp:
            self._fp.seek(pos, 0)
            if self._fp.readinto(memoryview(view)) != total:
                raise ValueError('file shorter than its seek table says')
            return
        if self._read_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._read_pool = ThreadPoolExecutor(max_workers=4)
        fd = self._fp.fileno()
        step = -(-total // nthr)

        def piece(i):
            lo, hi = i * step, min((i + 1) * step, total)
            got = 0
            while lo + got < hi:
                k = os.preadv(fd, [memoryview(view[lo + got:hi])], pos + lo + got)
                if k <= 0:
                    raise ValueError('file shorter than its seek table says')
                got += k
        list(self._read_pool.map(piece, range(nthr)))
        if move_fp:
            self._fp.seek(pos + total, 0)      # (where a plain read would have left the file)

    def seek_to_frame_data(self):
        self._frame_data_start_position = self._rc_header.get_frame_data_offset(self._is_intermediate,
                                                                                self._sz_frame_metadata)
        self._fp.seek(0, 2)
        if self._frame_data_start_position <= self._fp.tell():
            self._fp.seek(self._frame_data_start_position, 0)

    def get_file_position(self):
        return self._fp.tell()

    def copy_headers_to(self, target_fp, source_header_length):
        self._fp.seek(0, 0)
        target_fp.write(self._fp.read(self._rc_header.recode_header_length))
        target_fp.write(self._fp.read(source_header_length))

    # ---- frame access --------------------------------------------------------------------------------------
    def _pack(self, key, metadata, sparse):
        stats = None
        if isinstance(sparse, tuple):  # level 2: (binary map as COO, summary statistics)
            sparse, stats = sparse
        if sparse is None:
            self._header['nz'] = self._current_frame_index
            return None
        d = {'metadata': metadata, 'data': sparse}
        if stats is not None:
            d['summary_stats'] = stats
        return {key: d}

    def get_frame(self, z):
        if self._is_intermediate:
            raise ValueError("Random acceess is not available for intermediate files")
        if z >= self._header['nz']:
            raise ValueError('Requested frame index is greater than number of frames in dataset')
        coo = self._readahead_frame(int(z))
        if coo is not None:
            # the file position goes where the frame-at-a-time path would have left it: behind frame z (get_next_frame on a merged file
            # reads from there when the read-ahead has nothing for it - an empty frame, the last frame)
            self._fp.seek(self._frame_data_start_position + int(self._seek_table[z, 1]) + int(self._seek_table[z, 0]), 0)
            self._current_frame_index = z + 1
            return self._pack(z, self._frame_metadata[z], coo)
        self._fp.seek(self._frame_data_start_position + int(self._seek_table[z, 1]), 0)
        if self._file_size - self._fp.tell() == 0:
            return self._pack(z, None, None)
        out = self._pack(z, self._frame_metadata[z], self._get_frame_sparse(self._frame_metadata[z]))
        if out is not None:
            self._current_frame_index = z + 1
        return out

    def _next_header(self):
        """(frame_id, metadata) of the next frame, or None at the end of an intermediate file."""
        if self._current_frame_index == 0:
            self._fp.seek(self._frame_data_start_position, 0)
        if self._file_size - self._fp.tell() == 0:
            return None
        if not self._is_intermediate:
            if self._current_frame_index >= self._header['nz']:
                raise ValueError('Requested frame index is greater than number of frames in dataset')
            return self._current_frame_index, self._frame_metadata[self._current_frame_index]
        frame_id = np.frombuffer(self._fp.read(4), dtype=np.uint32)[0]
        return frame_id, self._read_metadata_row()

    def get_next_frame(self):
        z = self._current_frame_index
        # sequential by definition: from the third call on the frames come out of the batched reader (part files included, whose
        # records it indexes then) and this call only wraps its rows
        coo = self._readahead_frame(z)                    # (leaves the frame counter and the file position as they were)
        if coo is not None:
            # the file position goes where the frame-at-a-time path would have left it
            self._fp.seek(self._frame_data_start_position + int(self._seek_table[z, 1]) + int(self._seek_table[z, 0]), 0)
            self._current_frame_index = z + 1
            return self._pack(int(self.part_frame_ids[z]) if self._is_intermediate else z, self._frame_metadata[z], coo)
        nxt = self._next_header()
        if nxt is None:
            return None
        frame_id, d = nxt
        out = self._pack(frame_id, d, self._get_frame_sparse(d))
        if out is not None:
            self._current_frame_index += 1
        return out

    def get_next_frame_raw(self, read_data=True):
        nxt = self._next_header()
        if nxt is None:
            return None
        frame_id, d = nxt
        raw_d = self._get_frame_raw(d, read_data=read_data)
        if not read_data:
            raw_d = self._fp.tell()
        self._current_frame_index += 1
        return {frame_id: {'metadata': d, 'data': raw_d}}

    def _stream_sizes(self, md):
        """(bytes of the binary-map stream, bytes of the value stream or None) as stored in the file."""
        h = self._header
        level, mode = h['reduction_level'], h['rc_operation_mode']
        sz_map = self._structures.binary_image_sz_bytes if mode == 0 else int(md['bytes_in_compressed_binary_map'])
        if level not in (1, 2):
            return sz_map, None
        what = 'pixvals' if level == 1 else 'summary_stats'
        return sz_map, int(md['bytes_in_%s_%s' % ('packed' if mode == 0 else 'compressed', what)])

    def _get_frame_raw(self, frame_metadata, read_data=True):
        sz_map, sz_val = self._stream_sizes(frame_metadata)

        def take(n):
            if read_data:
                return self._fp.read(n)
            self._fp.seek(n, 1)
            return None
        out = {'binary_map': take(sz_map)}
        if sz_val is not None:
            out['pixvals'] = take(sz_val)
        return out

    def _get_frame_sparse_fused(self, md, sz_map, sz_val):
        h = self._header
        d, level = int(h['target_bit_depth']), int(h['reduction_level'])
        sz_val = sz_val if level == 1 else 0
        npk = int(md['bytes_in_packed_pixvals']) if level == 1 else 0
        blob = np.frombuffer(self._fp.read(sz_map + sz_val), np.uint8)
        if blob.size != sz_map + sz_val or blob.size == 0:
            return NotImplemented
        sizes = np.array([[sz_map, sz_val, npk]], np.uint32)
        prefix = np.zeros(2, np.uint64)
        args = (int(h['nx']), int(h['ny']), d, level, int(h['rc_operation_mode']), int(h['compression_scheme']), _lib.ptr(blob), _lib.ptr(sizes), 1)
        L = _lib.lib()
        if level == 1:
            cap = (npk * 8) // d       # the packed stream's length bounds the count: one call
        else:
            if L.rc_expand_frames(*args, _lib.ptr(prefix), None, 0) != _lib.RC_OK:     # bitmap only: a counting call first
                return NotImplemented
            cap = int(prefix[1])
        if cap == 0:
            return NotImplemented      # (an empty frame: the plain path knows the reference's conventions for it)
        coo = d <= 16                  # the matrix's own arrays straight from the device (rc_expand_frames_coo), else triplet rows
        dst = _BatchOut(coo).room(cap)
        st = dst.fn(L)(*args, _lib.ptr(prefix), dst.ptr(), cap)
        if st != _lib.RC_OK:
            if st == _lib.RC_ERR_UNSUPPORTED and int(h['rc_operation_mode']) == 1:
                self._foreign_file = True      # a stock encoder's streams: this file is not offered to the device decoders again
            return NotImplemented      # foreign or damaged: the stock decoder is the judge
        n = int(prefix[1])
        if n == 0:
            return NotImplemented
        if not coo:
            return self._make_coo_frame(n, dst.result(n))
        rows, cols, vals = dst.result(n)       # (views of this call's own buffer: the matrix keeps it alive)
        return self._coo_from_arrays(vals if vals.dtype == self._numpy_dtype else vals.astype(self._numpy_dtype), rows, cols)

    def _get_frame_sparse(self, frame_metadata):
        """Read one frame's streams, decompress if needed, expand on the GPU, wrap as COO (reference :379-471)."""
        h = self._header
        level, mode = h['reduction_level'], h['rc_operation_mode']
        sz_map, sz_val = self._stream_sizes(frame_metadata)
        if level in (1, 3) and (mode == 0 or h['compression_scheme'] in (1, 2)) and not self._no_fused_frame \
                and not (mode == 1 and self._foreign_file):
            # one device call for the whole frame (rc_expand_frames, n = 1): compressed streams in, triplets out - the decoded binary
            # map and value stream never visit the host (the reference's three steps below remain for everything it does not take)
            pos = self._fp.tell()
            coo = self._get_frame_sparse_fused(frame_metadata, sz_map, sz_val)
            if coo is not NotImplemented:
                return coo
            self._fp.seek(pos, 0)
        binary_map = self._fp.read(sz_map)
        values = self._fp.read(sz_val) if sz_val is not None else None
        if mode == 1:
            scheme = h['compression_scheme']
            # zstd / LZ4 streams that arrive here were refused by the fused call: a stock encoder's serial chains (DESIGN.md "Foreign
            # streams").  The stock library on the host walks one in a millisecond or two; the device's seam-2 LZ4 decoder would take
            # linked 64 KiB blocks too, but with ONE thread per frame (0.3 s per 4096 x 4096 binary map) - it stays what
            # recode_compressors.de_compress offers, not what the reader uses.
            host = compressors.host_stream_decoder(scheme) if scheme in (1, 2) and level in (1, 3) else None
            if host is not None:
                binary_map = host(binary_map, self._structures.binary_image_sz_bytes)
                if values is not None:
                    values = host(values, int(frame_metadata['bytes_in_packed_pixvals']) if level == 1 else 0)
            else:
                binary_map = compressors.de_compress(scheme, binary_map, self._decompressor_context)
                if values is not None:
                    values = compressors.de_compress(scheme, values, self._decompressor_context)
        # size the triplet buffer from the value stream (L1) or ask the library to count (bitmap-only levels)
        d = int(h['target_bit_depth'])
        if level == 1:
            cap = (len(values) * 8) // d
        else:
            cap = self._c_reader.count(binary_map)
        buf = np.empty((max(cap, 1), 3), dtype=np.uint64)
        n = self._c_reader.get_frame_sparse(level, binary_map, values if level == 1 else None, buf) if cap else 0
        if n == 0 and mode == 0:
            return (None, None) if level == 2 else None  # reference :387-391: an empty reduce-only frame reads as end of data
        coo = self._make_coo_frame(n, buf)
        if level != 2:
            return coo
        # level 2: the value stream holds one statistic per connected component, in scipy label order (reference :473-481,
        # whose count formula and unpacker are defective - SURVEY appendix B; the intended semantics are implemented)
        n_stats = (len(values) * 8) // d
        if d < 8 and n_stats and (8 * (len(values) - 1)) // d + 1 < n_stats:
            # fields narrower than a byte: the stream's length (all the record holds) fits more than one count - the padding behind the last
            # statistic can be as wide as a field.  The count is the number of 8-connected components of the binary map, as the writer's was.
            import scipy.ndimage as nd
            dense = np.zeros((int(h['ny']), int(h['nx'])), dtype=bool)
            dense[coo.row, coo.col] = True
            n_stats = min(n_stats, int(nd.label(dense, structure=np.ones((3, 3), dtype=int))[1]))
        stats = np.zeros(max(n_stats, 1), dtype=np.uint64)
        if n_stats:
            self._c_reader.bit_unpack_pixel_intensities(n_stats, values, stats)
        return coo, stats[:n_stats].astype(self._numpy_dtype)

    def _make_coo_frame(self, n, buf):
        """(row, col, value) uint64 triplets -> the COO matrix the reference returns (recode_reader.py:466-469).  The arrays are handed to
        an empty matrix instead of going through the constructor, whose index checks (min / max over both index arrays, dtype
        negotiation, copies) cost more than the device call that produced the triplets; rows and columns come from the expand kernel,
        inside the frame by construction."""
        d = buf[:n]
        dt = np.dtype(self._numpy_dtype)
        if dt.kind in 'ui' and d.flags['C_CONTIGUOUS']:
            # one pass over the rows (rc_split_triplets) instead of three strided conversions
            data, row, col = np.empty(n, dt), np.empty(n, np.int32), np.empty(n, np.int32)
            _lib.check(_lib.lib().rc_split_triplets(_lib.ptr(d), n, _lib.ptr(row), _lib.ptr(col), _lib.ptr(data), dt.itemsize), 'rc_split_triplets')
        else:
            data, row, col = d[:, 2].astype(dt), d[:, 0].astype(np.int32), d[:, 1].astype(np.int32)
        return self._coo_from_arrays(data, row, col)

    def _coo_from_arrays(self, data, row, col):
        m = coo_matrix((int(self._header['ny']), int(self._header['nx'])), dtype=self._numpy_dtype)
        if hasattr(m, 'coords'):
            m.data, m.coords = data, (row, col)
        else:
            m.data, m.row, m.col = data, row, col
        m.has_canonical_format = False          # (what the constructor leaves; the entries are in fact sorted and unique)
        return m


def merge_parts(folder_path, base_filename, num_parts):
    """Merge `<base>_partNNN` files into `<base>` (reference :495-595): header of part 000 (nz patched), the source header,
    one metadata row per frame (frame_id dropped), then the frame data blobs in frame-id order.  Two passes like the
    reference's, with bounded memory: the first reads the record headers only (frame data skipped), the second copies each
    frame's data from its part file to its place."""
    from .parallel import _copy_range, part_index
    paths = [os.path.join(folder_path, '%s_part%03d' % (base_filename, i)) for i in range(num_parts)]
    hdrs, tables, offsets = [], [], []
    for p in paths:
        h, rows, pos = part_index(p)
        hdrs.append(h)
        tables.append(rows)
        offsets.append(pos)
    rc_header = hdrs[0]
    header = rc_header.as_dict()
    n_md = max((t.shape[1] - 2 for t in tables if t.shape[0]), default=0)

    def stream(idx):
        for k in range(tables[idx].shape[0]):
            yield int(tables[idx][k, 0]), idx, k

    order = list(heapq.merge(*(stream(i) for i in range(num_parts))))
    head_len = rc_header.recode_header_length + int(header['source_header_length'])
    with open(os.path.join(folder_path, base_filename), 'wb') as target:
        with open(paths[0], 'rb') as src:
            target.write(src.read(head_len))
        for _, idx, k in order:
            target.write(np.ascontiguousarray(tables[idx][k, 2:2 + n_md], dtype='<u4').tobytes())
        target.flush()
        fds = [os.open(p, os.O_RDONLY) for p in paths]
        try:
            out_fd = target.fileno()
            dst = head_len + len(order) * 4 * n_md
            for _, idx, k in order:
                size = int(tables[idx][k, 1])
                _copy_range(fds[idx], out_fd, int(offsets[idx][k]), dst, size)
                dst += size
        finally:
            for fd in fds:
                os.close(fd)
        target.seek(rc_header.get_field_position_in_bytes('nz'), 0)
        target.write(len(order).to_bytes(rc_header.get_definition('nz')['bytes'], 'little'))
