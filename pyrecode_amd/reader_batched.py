"""The batched side of ReCoDeReader (no counterpart in the reference, which reads frame by frame): many frames per device call
(rc_expand_frames & co.), the streaming iterators, part files indexed for batched access, the host-decoded path for streams only a
stock decoder takes, and the read-ahead that serves the reference's own frame-at-a-time calls out of batches.  ReCoDeReader
(recode_reader.py) inherits BatchedAccess; the methods here use its file handle, header, metadata table and COO helpers.
"""
import os

import numpy as np

from . import recode_compressors as compressors
from . import _lib
from .misc import effective_cpus


def _split_wide(trip):
    """uint64 triplet rows -> the COO layout's three arrays with uint32 values (files whose values do not fit the device's uint16 COO form)."""
    trip = np.asarray(trip).reshape(-1, 3)
    return trip[:, 0].astype(np.int32), trip[:, 1].astype(np.int32), trip[:, 2].astype(np.uint32)


class _BatchOut:
    """Where a batch's expanded entries go and how they are laid out: the reference's uint64 (row, col, value) rows (24 bytes a set
    pixel; rc_expand_frames), or the three arrays of the COO matrix its reader wraps them into - int32 rows | int32 columns | uint16
    values, each `cap` entries long (10 bytes; rc_expand_frames_coo).  holder: None (an array of its own per call) or a one-element
    list with a _lib.PinnedBuffer / None (page-locked, reused, grown when too small: results are views, valid until its next use)."""

    def __init__(self, coo=False, holder=None):
        self.coo, self.holder, self.esz = bool(coo), holder, 10 if coo else 24
        self.buf, self.cap = None, 0

    def room(self, cap):
        nbytes = cap * self.esz
        if self.holder is None:
            self.buf = np.empty(nbytes, np.uint8)
        else:
            if self.holder[0] is None or self.holder[0].nbytes < nbytes:
                if self.holder[0] is not None:
                    self.holder[0].close()
                self.holder[0] = _lib.PinnedBuffer(max(int(nbytes * 1.25), 1 << 20))
            self.buf = self.holder[0].array[:nbytes]
        self.cap = cap
        return self

    def ptr(self):
        return _lib.ptr(self.buf)

    def fn(self, L, submit=False):
        if submit:
            return L.rc_expand_frames_coo_submit if self.coo else L.rc_expand_frames_submit
        return L.rc_expand_frames_coo if self.coo else L.rc_expand_frames

    def result(self, total):
        return self.views(self.buf, self.cap, total, self.coo)

    @staticmethod
    def views(buf, cap, total, coo):
        if not coo:
            return buf[:total * 24].view(np.uint64).reshape(total, 3)
        return (buf[:4 * cap].view(np.int32)[:total], buf[4 * cap:8 * cap].view(np.int32)[:total], buf[8 * cap:10 * cap].view(np.uint16)[:total])

    @staticmethod
    def from_triplets(trip, coo):
        """the frame-at-a-time path's triplets in the layout asked for"""
        if not coo:
            return trip
        return (trip[:, 0].astype(np.int32), trip[:, 1].astype(np.int32), trip[:, 2].astype(np.uint16))


class BatchedAccess:
    """Mixin of ReCoDeReader: see the module docstring.  State it uses is declared in ReCoDeReader.__init__."""

    def _load_part_index(self):
        """Intermediate (part) files carry no metadata table: every record is `u32 frame_id | metadata row | data` (reference
        recode_writer.py:559-574).  One walk over the record headers gives the batched readers what the seek table gives them for a
        merged file: per record the metadata row, the data's size and position - and the frame id, in `part_frame_ids`.  A record cut
        short at the end of the file (a writer that was interrupted) ends the index."""
        h = self._header
        level, mode = h['reduction_level'], h['rc_operation_mode']
        fields = self._md_fields()
        hdr = 4 + self._sz_frame_metadata
        std = sum(int(f['bytes']) for f in fields)
        fd = self._fp.fileno()
        pos, ids, mds, sizes, offs = self._frame_data_start_position, [], [], [], []
        while pos + hdr <= self._file_size:
            raw = os.pread(fd, hdr, pos)
            if len(raw) < hdr:
                break
            d, at = {}, 4
            for f in fields:
                d[f['name']] = np.frombuffer(raw, dtype=f['dtype'], count=1, offset=at)[0]
                at += int(f['bytes'])
            size = int(self._structures.get_frame_data_size(level, mode, d))
            if pos + hdr + size > self._file_size:
                break
            ids.append(int(np.frombuffer(raw, np.uint32, 1)[0]))
            mds.append(d)
            sizes.append(size)
            offs.append(pos + hdr - self._frame_data_start_position)
            pos += hdr + size
        assert std <= self._sz_frame_metadata
        self._frame_metadata = mds
        self.part_frame_ids = np.array(ids, np.uint32)
        self._seek_table = np.zeros((len(ids), 2), np.uint64)
        self._seek_table[:, 0] = sizes
        self._seek_table[:, 1] = offs            # (relative to the first record, like a merged file's - but NOT contiguous)

    def _batch_frames(self):
        """number of frames the batched readers can address: nz of a merged file, the records of a part file (indexed on first use)"""
        if self._is_intermediate:
            if self._frame_metadata is None:
                self._load_part_index()
            return len(self._frame_metadata)
        return int(self._header['nz'])

    def _read_batch_into(self, blob, z0, n, move_fp=True):
        """the data of frames z0 .. z0+n-1 -> blob, back to back (what rc_expand_frames takes): one contiguous range of a merged file;
        a part file's records have their headers in between, so frame by frame"""
        lo = self._frame_data_start_position + int(self._seek_table[z0, 1])
        if not self._is_intermediate:
            return self._read_into(blob, lo, move_fp)
        fd = self._fp.fileno()
        sizes = self._seek_table[z0:z0 + n, 0].astype(np.int64)
        ats = np.concatenate([[0], np.cumsum(sizes)])

        def some(lo, hi):
            for i in range(lo, hi):
                size, at = int(sizes[i]), int(ats[i])
                pos, got = self._frame_data_start_position + int(self._seek_table[z0 + i, 1]), 0
                while got < size:
                    k = os.preadv(fd, [memoryview(blob[at + got:at + size])], pos + got)
                    if k <= 0:
                        raise ValueError('file shorter than its records say')
                    got += k
        nthr = 4 if int(ats[-1]) >= (8 << 20) and n >= 4 else 1      # (as _read_into: a read from the page cache is a memcpy)
        if nthr == 1:
            return some(0, n)
        if self._read_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._read_pool = ThreadPoolExecutor(max_workers=4)
        cuts = [n * t // nthr for t in range(nthr + 1)]
        list(self._read_pool.map(lambda t: some(cuts[t], cuts[t + 1]), range(nthr)))

    _RA_FRAMES = 32          # frames fetched ahead once the calls turn out to be sequential
    _RA_BYTES = 512 << 20    # ... as long as their triplets fit into this much page-locked memory
    _RA_PIPELINED = True     # the batch behind the current one is prepared meanwhile (False: one synchronous batched call per window)

    def _readahead_frame(self, z):
        """The frame-at-a-time calls of the reference (get_frame in a loop, get_next_frame) served out of the batched reader: after
        three calls in sequence the frames from z on are fetched a batch at a time (get_frames_triplets into a page-locked buffer
        of this reader's), and the following calls only wrap their rows as COO.  Returns the COO matrix, or None for "take the
        frame-at-a-time path" (not sequential, level 2, an empty frame - whose conventions that path knows -, or a batch the batched
        reader could not deliver: the per-frame path then names the frame that is to blame).  The iterator stays alive between
        calls (the batch behind the current one is prepared meanwhile); one that finds a device slot taken by another reader's sends that
        batch through the synchronous call, so any number of readers in a process may do this side by side."""
        if self._ra_off or int(self._header['reduction_level']) not in (1, 3):
            return None
        if int(self._header['target_bit_depth']) > 16:
            return None        # (the read-ahead's batches come as the COO arrays with uint16 values: wider values keep the frame-at-a-time path)
        if self._user_iters:
            # a caller's own iter_frames_* on THIS reader is alive: it owns the reader's page-locked batch buffers (the read-ahead's
            # iterator would write the next batch into what that one's arrays view) - these calls go frame by frame meanwhile
            return None
        ra = self._ra
        fetch = False
        if ra is not None and not (ra[0] <= z < ra[0] + ra[1]):
            fetch = z == ra[0] + ra[1]                   # the batch behind the one just used up
            ra = self._ra = None
            if not fetch:
                self._close_ra_iter()
        if ra is None and not fetch:
            last = self._ra_last
            self._ra_streak = self._ra_streak + 1 if last is not None and z == last + 1 else 0
            fetch = self._ra_streak >= 2
        self._ra_last = z
        if ra is None:
            if not fetch:
                return None
            nz = self._batch_frames()
            if nz - z < 2:
                return None
            k, d, level = self._RA_FRAMES, int(self._header['target_bit_depth']), int(self._header['reduction_level'])
            # batches sized by what they expand to (10 bytes per set pixel): the value stream's length says how many there are; a
            # bitmap-only file does not - one set pixel in ten is assumed
            if level == 1:
                per = max(int(self._frame_metadata[z]['bytes_in_packed_pixvals']) * 8 // d * 10, 1)
            else:
                per = max(int(self._header['nx']) * int(self._header['ny']), 1)
            k = min(max(2, min(k, self._RA_BYTES // per)), nz - z)
            if self._ra_buf is None:
                self._ra_buf = [None]
            keep = (self._current_frame_index, self._fp.tell())
            mode, scheme = int(self._header['rc_operation_mode']), int(self._header['compression_scheme'])
            try:
                if (mode == 1 and (scheme in (0, 4, 5) or self._foreign_file)) or self._RA_PIPELINED:
                    # The streaming iterator, kept alive between calls: while the caller works through this batch the next one is on its
                    # way - decoded by the helper thread (streams only a stock decoder takes), or read, walked and queued on the device
                    # (this library's own streams; the iterator of another reader that finds the slot taken goes through the
                    # synchronous call for that batch, so readers side by side still all get their frames).
                    if self._ra_iter is None or self._ra_iter_at != z:
                        self._close_ra_iter()
                        self._ra_iter = self._iter_frames_impl(z, nz - z, batch=k, coo=True)
                    try:
                        a, prefix, arrays = next(self._ra_iter)
                    except StopIteration:
                        self._close_ra_iter()
                        return None
                    k = len(prefix) - 1
                    self._ra_iter_at = z + k
                else:
                    prefix, arrays = self.get_frames_triplets(z, k, out=self._ra_buf, coo=True)
            except Exception:
                self._close_ra_iter()
                self._ra_off = True
                return None
            finally:
                self._current_frame_index = keep[0]
                self._fp.seek(keep[1], 0)
            if self.last_batch_path == 'per-frame':      # nothing batched about this file: frame by frame it is
                self._ra_off = True
                return None
            ra = self._ra = (z, k, prefix, arrays)
        a, _, prefix, (rows, cols, vals) = ra
        lo, hi = int(prefix[z - a]), int(prefix[z - a + 1])
        if hi == lo:
            return None
        self.readahead_frames_served = self.readahead_frames_served + 1
        # the batch came as the COO arrays themselves (rc_expand_frames_coo): the frame's matrix takes its own copies of its slices
        return self._coo_from_arrays(vals[lo:hi].astype(self._numpy_dtype), rows[lo:hi].copy(), cols[lo:hi].copy())

    def _close_ra_iter(self):
        it, self._ra_iter, self._ra_iter_at = self._ra_iter, None, -1
        if it is not None:
            it.close()

    def _drop_readahead(self):
        self._ra = None
        self._close_ra_iter()
        buf = self._ra_buf
        if buf is not None and buf[0] is not None:
            buf[0].close()
        self._ra_buf = None

    # ---- batched access (device-resident decode + expand; no counterpart in the reference, which reads frame by frame) ------
    def _wide_values(self):
        """Level-1 files whose values are wider than 16 bits (target_bit_depth > 16: uint32 sources, recode_reader.py:56 / misc.py:41-49):
        rc_expand_frames_coo carries uint16 values (include/recode_hip.h), so the COO forms of the batched calls take such files through the
        24-byte triplets and split them on the host - same arrays, values as uint32."""
        h = self._header
        return int(h['reduction_level']) == 1 and int(h['target_bit_depth']) > 16

    def get_frames_triplets(self, z0, n, out=None, coo=False):
        """Frames z0 .. z0+n-1 of a merged file - or records z0 .. z0+n-1 of a part file, whose frame ids are part_frame_ids[z] - in ONE
        device call (rc_expand_frames): both streams of every frame are
        decompressed and expanded on the GPU without a host round trip in between.  Returns (nnz_prefix uint64[n+1],
        triplets uint64[total, 3]) - frame i's (row, col, value) rows are triplets[nnz_prefix[i]:nnz_prefix[i+1]], in the
        reference's row-major order (pyrecode.cpp:95-119).  Falls back to the per-frame path (stock decoder on the host) for
        streams outside the device decoders' subset, for level 2 and for host-only schemes.
        out: None (the triplets come in an array of their own), or a one-element list holding a _lib.PinnedBuffer or None - the
        triplets are then written into that page-locked buffer (grown when too small) and the returned array is a view of it,
        valid until the next call with the same holder.
        coo=True: instead of the triplet rows, (rows int32[total], columns int32[total], values uint16[total]) - the arrays of the COO
        matrices the frame-at-a-time calls return, 10 instead of 24 bytes per set pixel over the link (rc_expand_frames_coo)."""
        h = self._header
        if coo and self._wide_values():
            prefix, trip = self.get_frames_triplets(z0, n, out=out, coo=False)
            return prefix, _split_wide(trip)
        dst = _BatchOut(coo, out)
        nz = self._batch_frames()
        if z0 < 0 or n <= 0 or z0 + n > nz:
            raise ValueError('Requested frame index is greater than number of frames in dataset')
        level, mode, scheme = int(h['reduction_level']), int(h['rc_operation_mode']), int(h['compression_scheme'])
        fast = level in (1, 3) and (mode == 0 or scheme in (1, 2))
        host_only = level in (1, 3) and mode == 1 and scheme in (0, 4, 5)   # zlib / bz2 / lzma: stock decoder on the thread pool, ONE device expand
        if fast or host_only:
            sizes = np.zeros((n, 3), np.uint32)
            for i in range(n):
                md = self._frame_metadata[z0 + i]
                sz_map, sz_val = self._stream_sizes(md)
                sizes[i, 0] = sz_map
                if level == 1:
                    sizes[i, 1] = sz_val
                    sizes[i, 2] = int(md['bytes_in_packed_pixvals'])
            lo = self._frame_data_start_position + int(self._seek_table[z0, 1])
            total = int(self._seek_table[z0:z0 + n, 0].sum())
            if self._pin_blob is None or self._pin_blob.nbytes < total:   # file -> page-locked memory, no copy in between
                if self._pin_blob is not None:
                    self._pin_blob.close()
                self._pin_blob = _lib.PinnedBuffer(max(int(total * 1.25), 1 << 20))
            blob = self._pin_blob.array[:total]
            self._read_batch_into(blob, z0, n)
            prefix = np.zeros(n + 1, np.uint64)
            L = _lib.lib()
            args = (int(h['nx']), int(h['ny']), int(h['target_bit_depth']), level, mode, scheme, _lib.ptr(blob), _lib.ptr(sizes), n)
            if host_only or (mode == 1 and self._foreign_file):
                st = _lib.RC_ERR_UNSUPPORTED            # (a file whose streams the device decoders refused once is not offered again)
            elif level == 1:
                # a frame's packed stream holds one depth-bit field per set pixel: its size bounds the count, one call does it all
                d = int(h['target_bit_depth'])
                cap = max(int((sizes[:, 2].astype(np.uint64) * 8 // d).sum()), 1)
                st = dst.fn(L)(*args, _lib.ptr(prefix), dst.room(cap).ptr(), cap)
            else:
                st = L.rc_expand_frames(*args, _lib.ptr(prefix), None, 0)        # level 3: a counting call sizes the output
                if st == _lib.RC_OK:
                    cap = max(int(prefix[n]), 1)
                    st = dst.fn(L)(*args, _lib.ptr(prefix), dst.room(cap).ptr(), cap)
            if st == _lib.RC_OK:
                self._note_batch_end(z0 + n)
                self.last_batch_path = 'device'
                return prefix, dst.result(int(prefix[n]))
            # Outside the device decoders' subset - or a stream they could not make sense of (a foreign encoder's independent 64 KiB
            # LZ4 blocks look like that): the per-frame path below decodes with the stock library, which is also the judge of whether
            # the file is really damaged.
            if st not in (_lib.RC_ERR_UNSUPPORTED, _lib.RC_ERR_CORRUPT):
                _lib.check(st, 'rc_expand_frames')
            if mode == 1:
                res = self._foreign_batch_triplets(z0, n, blob, sizes, dst)
                if res is not None:
                    self._note_batch_end(z0 + n)
                    self.last_batch_path = 'host-decode + device-expand'
                    self._foreign_file = not host_only
                    return res
        # per-frame path
        self.last_batch_path = 'per-frame'
        parts, prefix = [], np.zeros(n + 1, np.uint64)
        keep = self._fp.tell()
        for i in range(n):
            self._fp.seek(self._frame_data_start_position + int(self._seek_table[z0 + i, 1]), 0)
            m = self._get_frame_sparse(self._frame_metadata[z0 + i])     # (not `coo`: that is the caller's layout flag, used below)
            m = m[0] if isinstance(m, tuple) else m
            t = np.stack([m.row.astype(np.uint64), m.col.astype(np.uint64), m.data.astype(np.uint64)], axis=1) if m is not None and m.nnz \
                else np.zeros((0, 3), np.uint64)
            parts.append(t)
            prefix[i + 1] = prefix[i] + t.shape[0]
        if self._is_intermediate:
            self._fp.seek(keep, 0)           # (get_next_frame's cursor)
        else:
            self._note_batch_end(z0 + n)
        return prefix, _BatchOut.from_triplets(np.concatenate(parts) if parts else np.zeros((0, 3), np.uint64), coo)

    def _note_batch_end(self, z):
        if not self._is_intermediate:        # (a part file's sequential cursor is its file position, which the batched readers leave alone)
            self._current_frame_index = z

    def _foreign_batch_triplets(self, z0, n, blob, sizes, dst=None):
        """Streams a FOREIGN encoder wrote (the reference's own files: lz4.frame with linked 64 KiB blocks, libzstd with 4-stream
        literals and real offsets) are serial chains of some 10^5 dependent steps per frame - the stock decoder on a CPU core
        walks one in about a millisecond, a GPU lane needs ~1 us per step ("Foreign streams").  So they are decoded
        by the SAME library calls the reference makes (recode_compressors.py:46-49), all 2 n streams of the batch at once on a
        thread pool (the libraries release the GIL), into the stored-pieces layout of a mode-0 file, and ONE device call expands
        them (rc_expand_frames, op_mode 0).  None: a stock decoder is not available or rejected a stream (the per-frame path
        then reports it)."""
        h = self._header
        level, scheme = int(h['reduction_level']), int(h['compression_scheme'])
        got = self._host_decode_batch(blob, sizes, n, 2)        # (an image of its own: the iterator's two may be in use between its steps)
        if got is None:
            return None
        pieces, sizes0 = got
        L = _lib.lib()
        prefix = np.zeros(n + 1, np.uint64)
        args = (int(h['nx']), int(h['ny']), int(h['target_bit_depth']), level, 0, scheme, _lib.ptr(pieces), _lib.ptr(sizes0), n)
        if level == 1:
            d = int(h['target_bit_depth'])
            cap = max(int((sizes0[:, 2].astype(np.uint64) * 8 // d).sum()), 1)
        else:
            _lib.check(L.rc_expand_frames(*args, _lib.ptr(prefix), None, 0), 'rc_expand_frames')
            cap = max(int(prefix[n]), 1)
        dst = dst if dst is not None else _BatchOut()
        _lib.check(dst.fn(L)(*args, _lib.ptr(prefix), dst.room(cap).ptr(), cap), 'rc_expand_frames')
        return prefix, dst.result(int(prefix[n]))

    def _host_decode_batch(self, blob, sizes, n, slot):
        """The 2 n streams of a batch (file bytes in `blob`, stream sizes in `sizes`) through the stock decoder of the file's scheme on the
        thread pool, each straight into its place of a stored-pieces image in page-locked memory (sizes are known beforehand: the
        binary map's nb bytes, the value stream's bytes_in_packed_pixvals).  Returns (pieces, sizes of a mode-0 batch), or None when
        there is no stock decoder for the scheme or it rejected a stream.  `slot` picks one of three images: two for the streaming
        iterator (a batch is decoded while the device still copies the previous one in), one for the synchronous call."""
        from concurrent.futures import ThreadPoolExecutor
        h = self._header
        level, scheme = int(h['reduction_level']), int(h['compression_scheme'])
        nb = self._structures.binary_image_sz_bytes
        dec = compressors.host_stream_decoder(scheme)
        if dec is None:
            return None
        spans, off, dst = [], 0, 0
        sizes0 = np.zeros((n, 3), np.uint32)
        for i in range(n):
            spans.append((off, int(sizes[i, 0]), dst, nb))
            off += int(sizes[i, 0])
            dst += nb
            sizes0[i, 0] = nb
            if level == 1:
                npk = int(sizes[i, 2])
                spans.append((off, int(sizes[i, 1]), dst, npk))
                off += int(sizes[i, 1])
                dst += npk
                sizes0[i, 1] = sizes0[i, 2] = npk
        if self._pin_pieces is None:
            self._pin_pieces = [None, None, None]
        buf = self._pin_pieces[slot]
        if buf is None or buf.nbytes < dst + 64:
            if buf is not None:
                buf.close()
            buf = self._pin_pieces[slot] = _lib.PinnedBuffer(int(dst * 1.25) + (1 << 20))
        pieces = buf.array[:dst]
        # zstd / LZ4 / zlib: the library's own worker threads make the stock library's calls (rc_host_decode_streams) - no
        # interpreter in the loop; the other schemes, or a host without those shared libraries: the Python-level decoders on a pool
        L = _lib.lib()
        if scheme in (0, 1, 2) and L.rc_host_decoder_available(scheme):
            table = np.array([(a, b, c, e) for a, b, c, e in spans], np.uint64).reshape(-1, 4)
            src = np.frombuffer(memoryview(blob), np.uint8)
            st = L.rc_host_decode_streams(scheme, _lib.ptr(src), _lib.ptr(pieces), _lib.ptr(table), table.shape[0], 0)
            if st == _lib.RC_OK:
                return pieces, sizes0
            if st != _lib.RC_ERR_UNSUPPORTED:
                return None                              # a stream the stock decoder rejects: the per-frame path names it
        view = memoryview(blob)

        def one(sp):
            src_off, src_n, at, want = sp
            if want:
                dec(view[src_off:src_off + src_n], want, pieces[at:at + want])
        try:
            if self._decode_pool is None:
                self._decode_pool = ThreadPoolExecutor(max_workers=min(16, effective_cpus()[0]))
            spans.sort(key=lambda sp: -sp[1])            # longest streams first: the pool's tail is then a short one
            list(self._decode_pool.map(one, spans))
        except Exception:
            return None
        return pieces, sizes0

    def _iter_host_decoded(self, z0, n, batch, coo=False):
        """iter_frames_triplets for files whose streams only a stock decoder takes (foreign encoders, zlib / bz2 / lzma): batch i + 1 is
        read and decoded on the host's thread pool while the device expands batch i (rc_expand_frames_submit / _wait, op_mode 0, on the
        page-locked stored-pieces image) and the consumer works on its triplets."""
        from concurrent.futures import ThreadPoolExecutor
        h = self._header
        level, scheme, d = int(h['reduction_level']), int(h['compression_scheme']), int(h['target_bit_depth'])
        L = _lib.lib()
        geom0 = (int(h['nx']), int(h['ny']), d, level, 0, scheme)
        starts = list(range(z0, z0 + n, batch))
        if self._stream_bufs is None:
            self._stream_bufs = [None, None, None, None]
        if self._host_blobs is None:
            self._host_blobs = [None, None]
        if self._decode_coord is None:
            self._decode_coord = ThreadPoolExecutor(max_workers=1)
        if self._file_map is None:
            import mmap
            try:
                self._file_map = mmap.mmap(self._fp.fileno(), 0, access=mmap.ACCESS_READ)
            except (OSError, ValueError):
                self._file_map = None                                          # (not mappable: positional reads into a buffer instead)
        bufs = self._stream_bufs

        def prepare(i):
            a = starts[i]
            k = min(batch, z0 + n - a)
            slot = i & 1
            sizes = np.zeros((k, 3), np.uint32)
            for j in range(k):
                md = self._frame_metadata[a + j]
                sz_map, sz_val = self._stream_sizes(md)
                sizes[j, 0] = sz_map
                if level == 1:
                    sizes[j, 1], sizes[j, 2] = sz_val, int(md['bytes_in_packed_pixvals'])
            total = int(self._seek_table[a:a + k, 0].sum())
            lo = self._frame_data_start_position + int(self._seek_table[a, 1])
            if self._file_map is not None and not self._is_intermediate:
                blob = np.frombuffer(self._file_map, np.uint8, total, lo)      # the decoders read the page cache itself
            else:
                if self._host_blobs[slot] is None or self._host_blobs[slot].size < total:
                    self._host_blobs[slot] = np.empty(int(total * 1.25) + 64, np.uint8)
                blob = self._host_blobs[slot][:total]
                self._read_batch_into(blob, a, k, move_fp=False)
            return a, k, slot, self._host_decode_batch(blob, sizes, k, slot)

        fut = self._decode_coord.submit(prepare, 0) if starts else None
        submitted = None
        try:
            for i in range(len(starts)):
                a, k, slot, got = fut.result()
                fut = None
                if got is None:
                    # the stock decoder rejected a stream: the synchronous call goes frame by frame and names it (nothing decodes ahead
                    # meanwhile: that call reads the same file and may use the same pools)
                    res = (a,) + self.get_frames_triplets(a, k, coo=coo)
                fut = self._decode_coord.submit(prepare, i + 1) if i + 1 < len(starts) else None
                if got is None:
                    pass
                elif level != 1:
                    # (level 3 needs a counting call to size its output: the synchronous form does both)
                    pieces, sizes0 = got
                    prefix = np.zeros(k + 1, np.uint64)
                    args = geom0 + (_lib.ptr(pieces), _lib.ptr(sizes0), k)
                    _lib.check(L.rc_expand_frames(*args, _lib.ptr(prefix), None, 0), 'rc_expand_frames')
                    dst = _BatchOut(coo).room(max(int(prefix[k]), 1))
                    _lib.check(dst.fn(L)(*args, _lib.ptr(prefix), dst.ptr(), dst.cap), 'rc_expand_frames')
                    self.last_batch_path = 'host-decode + device-expand'
                    res = (a, prefix, dst.result(int(prefix[k])))
                else:
                    pieces, sizes0 = got
                    cap = max(int((sizes0[:, 2].astype(np.uint64) * 8 // d).sum()), 1)
                    esz = 10 if coo else 24
                    if bufs[2 + slot] is None or bufs[2 + slot].nbytes < cap * esz:
                        if bufs[2 + slot] is not None:
                            bufs[2 + slot].close()
                        bufs[2 + slot] = _lib.PinnedBuffer(max(int(cap * esz * 1.25), 1 << 20))
                    prefix = np.zeros(k + 1, np.uint64)
                    st = _BatchOut(coo).fn(L, submit=True)(slot, *geom0, _lib.ptr(pieces), _lib.ptr(sizes0), k, bufs[2 + slot]._p, cap)
                    if st == _lib.RC_ERR_BAD_ARG and 'submitted batch' in _lib.last_error():
                        # another iterator of this process holds the slot: the synchronous call has resources of its own
                        _lib.check(_BatchOut(coo).fn(L)(*geom0, _lib.ptr(pieces), _lib.ptr(sizes0), k, _lib.ptr(prefix), bufs[2 + slot]._p, cap),
                                   'rc_expand_frames')
                    else:
                        _lib.check(st, 'rc_expand_frames_submit')
                        submitted = (slot, k)
                        st = L.rc_expand_frames_wait(slot, _lib.ptr(prefix))
                        submitted = None
                        _lib.check(st, 'rc_expand_frames_wait')
                    total = int(prefix[k])
                    self.last_batch_path = 'host-decode + device-expand'
                    res = (a, prefix, _BatchOut.views(bufs[2 + slot].array, cap, total, coo))
                self._note_batch_end(a + k)
                yield res
        finally:
            if fut is not None:
                try:
                    fut.result()          # the decode that runs ahead writes into buffers close() frees
                except Exception:
                    pass
            if submitted is not None:
                try:
                    L.rc_expand_frames_wait(submitted[0], _lib.ptr(np.zeros(submitted[1] + 1, np.uint64)))
                except Exception:         # (a generator finalised while the interpreter shuts down)
                    pass

    def iter_frames_triplets(self, z0=0, n=None, batch=64, coo=False):
        """The caller's own streaming iterator (documented at _iter_frames_impl).  The read-ahead under get_frame / get_next_frame keeps
        an iterator of its own alive on the same page-locked buffers, with one batch queued on the device: it is ended first (its queued
        batch waited for, the window it serves forgotten), and it stays off while this generator lives."""
        self._ra = None
        self._close_ra_iter()
        self._user_iters += 1
        try:
            yield from self._iter_frames_impl(z0, n, batch, coo)
        finally:
            self._user_iters -= 1

    def _iter_frames_impl(self, z0=0, n=None, batch=64, coo=False):
        """Streams frames z0 .. z0+n-1 of a merged file (records z0 .. of a part file: the reference's own read test sums a part file's
        frames one get_next_frame at a time, tests/recode_v1_read_test.py:9-21) through the batched device reader, two batches in flight
        (rc_expand_frames_submit / _wait): while the device decodes one batch, the next one is read from the file, its block headers
        are walked and its bytes copied in.  Yields (first frame index, nnz_prefix uint64[k+1], triplets uint64[total, 3]) per batch
        of k <= `batch` frames; `triplets` is a VIEW of page-locked memory the device wrote directly - valid until the generator is
        advanced (copy it to keep it).  Files the device path does not take (level 2, host-only schemes, foreign streams) go through
        get_frames_triplets batch by batch.  coo=True: the third item is (rows int32, columns int32, values uint16) instead of the
        triplet rows - 10 instead of 24 bytes per set pixel over the link (rc_expand_frames_coo_submit)."""
        h = self._header
        if coo and self._wide_values():   # values beyond uint16: the device's COO layout does not hold them - triplets, split on the host
            for a, prefix, trip in self._iter_frames_impl(z0, n, batch, coo=False):
                yield a, prefix, _split_wide(trip)
            return
        nz = self._batch_frames()
        n = nz - z0 if n is None else n
        if z0 < 0 or n < 0 or z0 + n > nz or batch <= 0:
            raise ValueError('Requested frame index is greater than number of frames in dataset')
        level, mode, scheme = int(h['reduction_level']), int(h['rc_operation_mode']), int(h['compression_scheme'])
        d = int(h['target_bit_depth'])
        starts = list(range(z0, z0 + n, batch))
        if level in (1, 3) and mode == 1 and (scheme in (0, 4, 5) or (scheme in (1, 2) and self._foreign_file)):
            yield from self._iter_host_decoded(z0, n, batch, coo)  # stock decoders on the pool, one batch ahead of the device
            return
        if not (level == 1 and (mode == 0 or scheme in (1, 2))):
            for a in starts:
                k = min(batch, z0 + n - a)
                yield (a,) + self.get_frames_triplets(a, k, coo=coo)
            return
        L = _lib.lib()
        geom = (int(h['nx']), int(h['ny']), d, level, mode, scheme)
        if self._stream_bufs is None:
            self._stream_bufs = [None, None, None, None]       # page-locked: two input blobs, two outputs; kept until close()
        bufs = self._stream_bufs

        def pinned(buf, nbytes):
            if buf is None or buf.nbytes < nbytes:
                if buf is not None:
                    buf.close()
                buf = _lib.PinnedBuffer(max(int(nbytes * 1.25), 1 << 20))
            return buf

        def submit(i):
            """read batch i's bytes into its slot's page-locked blob and queue it; returns what wait needs, or None for 'not on the device'"""
            a = starts[i]
            k = min(batch, z0 + n - a)
            slot = i & 1
            sizes = np.zeros((k, 3), np.uint32)
            for j in range(k):
                md = self._frame_metadata[a + j]
                sizes[j, 0], sizes[j, 1] = self._stream_sizes(md)
                sizes[j, 2] = int(md['bytes_in_packed_pixvals'])
            total = int(self._seek_table[a:a + k, 0].sum())
            bufs[slot] = pinned(bufs[slot], total + 64)
            blob = bufs[slot].array[:total]
            self._read_batch_into(blob, a, k)
            cap = max(int((sizes[:, 2].astype(np.uint64) * 8 // d).sum()), 1)
            bufs[2 + slot] = pinned(bufs[2 + slot], cap * (10 if coo else 24))
            st = _BatchOut(coo).fn(L, submit=True)(slot, *geom, _lib.ptr(blob), _lib.ptr(sizes), k, bufs[2 + slot]._p, cap)
            if st in (_lib.RC_ERR_UNSUPPORTED, _lib.RC_ERR_CORRUPT):
                return (a, k, slot, None)
            if st == _lib.RC_ERR_BAD_ARG and 'submitted batch' in _lib.last_error():
                # the library's two streaming slots are per process and device: ANOTHER iterator (another reader) holds this one.
                # This batch goes through the synchronous call, which has resources of its own.
                return (a, k, slot, None)
            _lib.check(st, 'rc_expand_frames_submit')
            return (a, k, slot, cap)

        def finish(job):
            a, k, slot, cap = job
            if cap is None:
                return (a,) + self.get_frames_triplets(a, k, coo=coo)       # (sets last_batch_path itself)
            prefix = np.zeros(k + 1, np.uint64)
            st = L.rc_expand_frames_wait(slot, _lib.ptr(prefix))
            if st == _lib.RC_ERR_CORRUPT:                     # the stock decoder is the judge
                return (a,) + self.get_frames_triplets(a, k, coo=coo)
            _lib.check(st, 'rc_expand_frames_wait')
            total = int(prefix[k])
            trip = _BatchOut.views(bufs[2 + slot].array, cap, total, coo)
            self.last_batch_path = 'device'
            return a, prefix, trip
        queued = None        # a batch submitted and not yet waited for
        try:
            queued = submit(0) if starts else None
            for i in range(len(starts)):
                job = queued
                if self._foreign_file:
                    # the previous batch turned out to be a foreign encoder's: the rest of the file goes through the host-decoded pipeline
                    if job[3] is not None:
                        L.rc_expand_frames_wait(job[2], _lib.ptr(np.zeros(job[1] + 1, np.uint64)))
                    queued = None
                    yield from self._iter_host_decoded(job[0], z0 + n - job[0], batch, coo)
                    return
                queued = submit(i + 1) if i + 1 < len(starts) else None
                res = finish(job)
                self._note_batch_end(job[0] + job[1])
                yield res
        finally:
            # a consumer that stops early leaves a batch queued: wait for it before its buffers go away
            if queued is not None and queued[3] is not None:
                try:
                    L.rc_expand_frames_wait(queued[2], _lib.ptr(np.zeros(queued[1] + 1, np.uint64)))
                except Exception:         # (a generator finalised while the interpreter shuts down)
                    pass

    def get_frames_coo(self, z0, n, out=None):
        """get_frames_triplets in the COO layout: (nnz_prefix, (rows int32, columns int32, values uint16))"""
        return self.get_frames_triplets(z0, n, out=out, coo=True)

    def iter_frames_coo(self, z0=0, n=None, batch=64):
        """iter_frames_triplets in the COO layout: yields (first frame, nnz_prefix, (rows int32, columns int32, values uint16))"""
        return self.iter_frames_triplets(z0, n, batch, coo=True)

    def get_frames(self, z0, n):
        """{frame index: {'metadata', 'data': COO}} for n consecutive frames, decoded in one device call."""
        coo_ok = int(self._header['reduction_level']) in (1, 3) and int(self._header['target_bit_depth']) <= 16
        prefix, got = self.get_frames_triplets(z0, n, coo=coo_ok)
        out = {}
        for i in range(n):
            lo, hi = int(prefix[i]), int(prefix[i + 1])
            if coo_ok:      # the batch came as the matrices' own arrays: a frame's matrix takes copies of its slices
                coo = self._coo_from_arrays(got[2][lo:hi].astype(self._numpy_dtype), got[0][lo:hi].copy(), got[1][lo:hi].copy())
            else:
                coo = self._make_coo_frame(hi - lo, got[lo:hi])
            key = int(self.part_frame_ids[z0 + i]) if self._is_intermediate else z0 + i      # (what get_next_frame keys a part file's frames by)
            out[key] = {'metadata': self._frame_metadata[z0 + i], 'data': coo}
        return out
