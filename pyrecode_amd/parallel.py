"""Multi-GPU driver pieces: frame ownership, and the merged `.rcN` file written directly after ONE collective.

The reference scales by data parallelism over frames: node i owns the contiguous block [i*ceil(n/T), ...)
(pyrecode/recode_writer.py:320-322), every node writes its own part file, and the final file is produced afterwards by
merge_parts' two passes over all part files (pyrecode/recode_reader.py:495-595).  Here one process drives one GPU
(rank == node_id), frames never cross GPUs, and the merge needs a single exchange step: an all-gather of
the per-frame metadata rows (<= 12 bytes per frame; RCCL over xGMI when the process group is `nccl`, gloo on CPU).
From the gathered table every rank derives the global seek table and its own byte offset, and pwrite()s its frame data
into place.  The result is byte-identical to merge_parts on the same part files (tests/test_parallel_gloo.py).
"""
import os
import struct

import numpy as np

from .recode_reader import ReCoDeReader


def frame_block(n_frames, world, rank):
    """(first frame, frame count) owned by `rank` - the reference's contiguous-block rule (recode_writer.py:320-322)."""
    per = -(-n_frames // world)
    lo = rank * per
    return lo, min(per, max(n_frames - lo, 0))


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def all_gather_rows(rows, device_id=None):
    """rows: int64 ndarray [k, w] (k may differ per rank).  Returns the list of every rank's rows, in rank order.
    One collective for the payload (padded to the largest k) preceded by a tiny one for the counts.
    With RCCL the tensors live on `device_id` (default: the caller's current device); a rank whose current device is not the
    one its writer used would make two ranks meet on one GPU inside the collective, so that is refused here."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return [rows]
    import torch
    world = dist.get_world_size()
    on_gpu = dist.get_backend() == 'nccl'
    if on_gpu:
        cur = torch.cuda.current_device()
        if device_id is not None and int(device_id) != cur:
            raise RuntimeError('rank %d: current device is cuda:%d but its frames were reduced on cuda:%d; call '
                               'torch.cuda.set_device(local_rank) before the collective' % (dist.get_rank(), cur, int(device_id)))
        dev = torch.device('cuda', cur)
    else:
        dev = torch.device('cpu')
    w = rows.shape[1]
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev))
    counts = [int(c.item()) for c in counts]
    kmax = max(max(counts), 1)
    mine = torch.zeros((kmax, w), dtype=torch.int64, device=dev)
    if rows.shape[0]:
        mine[:rows.shape[0]] = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(dev)
    gathered = torch.empty((world * kmax, w), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, mine)
    g = gathered.cpu().numpy().reshape(world, kmax, w)
    return [g[r, :counts[r]] for r in range(world)]


def read_part_records(path):
    """[(frame_id, [metadata values], data bytes)] of one part file, in file order."""
    rd = ReCoDeReader(path, is_intermediate=True)
    rd.open(print_header=False)
    out = []
    while True:
        f = rd.get_next_frame_raw()
        if f is None:
            break
        (fid, body), = f.items()
        out.append((int(fid), [int(v) for v in body['metadata'].values()], b''.join(body['data'].values())))
    hdr = rd.get_header()
    rd.close()
    return hdr, out


def _copy_range(src_fd, dst_fd, src_off, dst_off, count, bufsize=32 << 20):
    """count bytes from src_fd@src_off to dst_fd@dst_off with bounded memory (copy_file_range where the kernel has it)."""
    while count > 0:
        n = 0
        if hasattr(os, 'copy_file_range'):
            try:
                n = os.copy_file_range(src_fd, dst_fd, min(count, 1 << 30), src_off, dst_off)
            except OSError:
                n = 0
        if n <= 0:
            buf = os.pread(src_fd, min(count, bufsize), src_off)
            if not buf:
                raise IOError('part file shorter than its frame index says')
            n = os.pwrite(dst_fd, buf, dst_off)
        src_off += n
        dst_off += n
        count -= n


def part_index(path):
    """(header, rows int64[n, 2 + n_md] = [frame_id, data bytes, metadata...], data offsets int64[n]) of one part file, read
    WITHOUT its frame data (headers only): what merge_direct needs when the writer did not hand its own index over."""
    rd = ReCoDeReader(path, is_intermediate=True)
    rd.open(print_header=False)
    rows, pos = [], []
    while True:
        f = rd.get_next_frame_raw(read_data=False)
        if f is None:
            break
        (fid, body), = f.items()
        md = [int(v) for v in body['metadata'].values()]
        end = int(body['data'])          # file position behind the frame's data
        h = rd.get_header().as_dict()
        size = rd._structures.get_frame_data_size(h['reduction_level'], h['rc_operation_mode'], body['metadata'])
        rows.append([int(fid), size] + md)
        pos.append(end - size)
    hdr = rd.get_header()
    rd.close()
    n_md = len(rows[0]) - 2 if rows else 0
    return hdr, np.array(rows, dtype=np.int64).reshape(len(rows), 2 + n_md), np.array(pos, dtype=np.int64)


def merge_direct(folder_path, base_filename, rank=None, world=None, records=None, index=None, device_id=None):
    """Collective.  Every rank contributes its own frames to `<base>`; rank 0 also writes the header (copy of part 000's, nz
    patched) and the metadata table.  A rank describes its frames by, in order of preference: `index` = (rows, offsets)
    from ReCoDeWriter.frame_index() (nothing is read back), the headers of its part file (part_index: frame data skipped),
    or `records` = [(frame_id, [metadata], data bytes)] held in memory.  Frame data is copied part file -> merged file
    with bounded memory.  Layout: reference recode_reader.py:518-592 / SURVEY appendix A."""
    dist = _dist()
    if rank is None:
        rank = dist.get_rank() if dist else 0
    if world is None:
        world = dist.get_world_size() if dist else 1
    part = os.path.join(folder_path, '%s_part%03d' % (base_filename, rank))
    pos = None
    if records is not None:
        rd = ReCoDeReader(part, is_intermediate=True)
        rd.open(print_header=False)
        hdr = rd.get_header()
        rd.close()
        n_md = len(records[0][1]) if records else 0
        rows = np.array([[fid, len(data)] + md for fid, md, data in records], dtype=np.int64).reshape(len(records), 2 + n_md)
    elif index is not None:
        rd = ReCoDeReader(part, is_intermediate=True)
        rd.open(print_header=False)
        hdr = rd.get_header()
        rd.close()
        rows, pos = index
    else:
        hdr, rows, pos = part_index(part)
    if rows.shape[0] == 0:  # width must agree across ranks: derive it from the header's (level, mode)
        from .structures import ReCoDeStructures
        h = hdr.as_dict()
        n_md = len(ReCoDeStructures(h).standard_frame_metadata_structure_for(h['reduction_level'], h['rc_operation_mode']))
        rows = np.zeros((0, 2 + n_md), dtype=np.int64)
    n_md = rows.shape[1] - 2
    per_rank = all_gather_rows(rows, device_id=device_id)
    table = np.concatenate(per_rank, axis=0)
    order = np.argsort(table[:, 0], kind='stable')          # frame-id order (already sorted for contiguous blocks)
    sizes = table[order, 1]
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]) if len(sizes) else np.zeros(0, np.int64)
    pos_of = dict(zip(table[order, 0].tolist(), starts.tolist()))
    nz = table.shape[0]
    h = hdr.as_dict()
    head_len = hdr.recode_header_length + int(h['source_header_length'])
    data_start = head_len + nz * 4 * n_md
    target = os.path.join(folder_path, base_filename)
    if rank == 0:
        with open(os.path.join(folder_path, '%s_part%03d' % (base_filename, 0)), 'rb') as src, open(target, 'wb') as out:
            out.write(src.read(head_len))
            out.write(np.ascontiguousarray(table[order, 2:], dtype='<u4').tobytes())
            out.seek(hdr.get_field_position_in_bytes('nz'))
            out.write(int(nz).to_bytes(hdr.get_definition('nz')['bytes'], 'little'))
            out.truncate(data_start + int(sizes.sum()))
    if dist is not None and world > 1:
        dist.barrier()                                      # the file exists and is sized before anyone pwrite()s
    fd = os.open(target, os.O_WRONLY)
    try:
        if records is not None:
            for fid, md, data in records:
                os.pwrite(fd, data, data_start + pos_of[fid])
        elif rows.shape[0]:
            src_fd = os.open(part, os.O_RDONLY)
            try:
                # consecutive frames whose data is consecutive in both files go in one copy (mode-0 records with no
                # metadata; otherwise every record's header sits in between and frames are copied one by one)
                i, n = 0, rows.shape[0]
                while i < n:
                    s_off, d_off, cnt = int(pos[i]), data_start + pos_of[int(rows[i, 0])], int(rows[i, 1])
                    j = i + 1
                    while j < n and int(pos[j]) == s_off + cnt and data_start + pos_of[int(rows[j, 0])] == d_off + cnt:
                        cnt += int(rows[j, 1])
                        j += 1
                    _copy_range(src_fd, fd, s_off, d_off, cnt)
                    i = j
            finally:
                os.close(src_fd)
    finally:
        os.close(fd)
    if dist is not None and world > 1:
        dist.barrier()
    return nz


def write_sharded(image_filename, data, dark_data, output_directory, input_params, validation_frame_gap=-1, batch_size=None,
                  device_id=None, merge=True):
    """One call per rank (rank == node_id): reduce-compress this rank's contiguous block of `data` on its GPU into
    `<stem>.rc<L>_part<rank>` with ReCoDeWriter (the reference's per-node flow, recode_server.py:688-723), then build the
    merged `<stem>.rc<L>` collectively with merge_direct.  `input_params.num_threads` must equal the world size.
    Returns (run_metrics of this rank, total frames in the merged file or None)."""
    from pathlib import Path

    from .recode_writer import ReCoDeWriter
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    if int(input_params.num_threads) != world:
        raise ValueError('input_params.num_threads (%s) must equal the number of ranks (%d)' % (input_params.num_threads, world))
    w = ReCoDeWriter(image_filename, dark_data=dark_data, output_directory=output_directory, input_params=input_params,
                     mode='batch', validation_frame_gap=validation_frame_gap, node_id=rank, batch_size=batch_size,
                     device_id=device_id)
    w.start()
    used_device = w.device_id()
    if dist is not None and world > 1 and dist.get_backend() == 'nccl':
        import torch
        torch.cuda.set_device(used_device)   # the collective's tensors must live on the GPU this rank's frames were reduced on
    metrics = w.run(data)
    index = w.frame_index()   # what this rank wrote, and where: the merge reads nothing back but the frame data it copies
    w.close()
    if dist is not None and world > 1:
        dist.barrier()  # every part file is complete before anyone reads part 000's header
    nz = None
    if merge:
        base = '%s.rc%d' % (Path(image_filename).stem, int(input_params.reduction_level))
        nz = merge_direct(output_directory, base, rank=rank, world=world, index=index, device_id=used_device)
    return metrics, nz


# ---- the weak-scaling step loop (bench.py --gpus N; tests/test_parallel_gloo.py drives it with two gloo ranks) -----------
class _NullStream:
    """Stand-in for a HIP stream where there is no device (CPU rehearsal over gloo): every operation completes at once."""
    cuda_stream = 0

    def wait_event(self, ev):
        pass


class _NullEvent:
    def record(self, stream=None):
        pass


class ShardedStepLoop:
    """One rank of the data-parallel hot path as the reference runs it (one writer per node, recode_server.py:350-363;
    contiguous frame blocks, recode_writer.py:320-322): every step reduces `batch` frames on this rank's GPU; the path's one
    exchange step, the all-gather of the per-frame metadata rows, is issued every `gather_every` steps on a SIDE
    stream so that it runs under the following steps' reduce kernels.

    gather_every = G >= 1: the rows of G consecutive steps form a group; the step that completes a group gathers it (G = 1: one
        collective per step, the form of rounds 1-3).  Groups are double-buffered: group g writes md2[g & 1] and is gathered
        into md_all2[g & 1]; before group g + 2 rewrites md2[g & 1] the main stream waits for that gather's event.
    gather_every = 0: ONE gather per fenced region - `fence()` gathers whatever the region's steps left (BASELINE's "RCCL only for
        the final merged-index gather", literally); `region_steps` bounds the steps between two fences.
    The two events are created once and re-recorded (an event object per step cost a hipEventCreate each).

    ctx          the device boundary: enqueue(frames_ptr, n, first_frame_id, out_ptr, out_cap, rec_ptr, md_ptr) and
                 wait_results(stream_handle) (pyrecode_amd._lib.ReduceContext; the CPU test passes a stand-in)
    frames_of    step index -> (address of this rank's `batch` frames, their first frame id)
    device       torch.device: cuda:<local rank> (HIP streams, RCCL) or cpu (no streams, gloo)
    """

    def __init__(self, ctx, batch, frames_of, out, rec, device, collective=True, gather_every=1, region_steps=1, fence_barrier=False):
        import torch
        self._torch = torch
        self.ctx, self.B, self.frames_of, self.device = ctx, int(batch), frames_of, device
        # out / rec: one buffer, or a list of two that consecutive steps alternate between (a pipelined ctx may still be writing step i's
        # records when step i + 1's second stage starts: include/recode_hip.h, rc_ctx_set_pipelined)
        self.outs = list(out) if isinstance(out, (list, tuple)) else [out]
        self.recs = list(rec) if isinstance(rec, (list, tuple)) else [rec]
        if len(self.outs) != len(self.recs):
            raise ValueError('as many rec_offsets buffers as output buffers')
        self._out_ptr = [(o.data_ptr(), o.numel()) for o in self.outs]
        self._rec_ptr = [r.data_ptr() for r in self.recs]
        self._enq = 0                     # steps enqueued so far: step n writes buffer n % len(outs)
        self.dist = _dist() if collective else None
        # collective=False, fence_barrier=True: no exchange step, but the ranks still meet at the fences (a rehearsal of N host-side
        # step loops whose timed regions must start and end together)
        self._fence_dist = self.dist if self.dist else (_dist() if fence_barrier else None)
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.on_gpu = device.type == 'cuda'
        self.gather_every = int(gather_every)
        if self.gather_every < 0:
            raise ValueError('gather_every must be >= 0')
        self.G = self.gather_every if self.gather_every else max(1, int(region_steps))   # steps a group's buffer holds
        if self.on_gpu:
            if torch.cuda.current_device() != device.index:
                raise RuntimeError('rank %d: current device cuda:%d is not the loop\'s device %s' % (self.rank, torch.cuda.current_device(), device))
            if getattr(ctx, 'device_id', device.index) != device.index:
                raise RuntimeError('rank %d: ctx lives on cuda:%s, loop on %s' % (self.rank, ctx.device_id, device))
            self.stream = torch.cuda.Stream(device=device)
            self.cstream = torch.cuda.Stream(device=device) if self.dist else None
            self._events = [torch.cuda.Event(), torch.cuda.Event()] if self.dist else None
        else:
            self.stream, self.cstream = _NullStream(), _NullStream()
            self._events = [_NullEvent(), _NullEvent()] if self.dist else None
        rows = self.G * self.B
        self.md2 = [torch.zeros((rows, 3), dtype=torch.int32, device=device) for _ in range(2)]
        self.md_all2 = [torch.zeros((self.world * rows, 3), dtype=torch.int32, device=device) for _ in range(2)] if self.dist else None
        self.gathered = [None, None]      # the event of the last gather of each buffer (None: never gathered)
        self._md_ptr = [[m[j * self.B].data_ptr() for j in range(self.G)] for m in self.md2]   # (no tensor indexing in the step loop)
        self._group, self._pos = 0, 0     # current group, steps it holds so far
        self.last_gathered = None         # (buffer index, steps in it) of the most recent gather
        self.gathers_issued = 0
        self.steps_done = 0

    def _side(self):
        import contextlib
        return self._torch.cuda.stream(self.cstream) if self.on_gpu else contextlib.nullcontext()

    def _gather_group(self):
        """the current group's rows -> every rank (side stream), then the next group begins"""
        k = self._group & 1
        if self.dist and self._pos:
            with self._side():
                self.ctx.wait_results(self.cstream.cuda_stream)   # the collective's stream waits for the group's last batch's metadata rows
                self.dist.all_gather_into_tensor(self.md_all2[k], self.md2[k])
                self._events[k].record(self.cstream)
                self.gathered[k] = self._events[k]
            self.last_gathered = (k, self._pos)
            self.gathers_issued += 1
        self._group += 1
        self._pos = 0

    def step(self, i):
        k = self._group & 1
        if self._pos == 0 and self.dist and self.gathered[k] is not None:
            self.stream.wait_event(self.gathered[k])   # md2[k] is rewritten from here on: its previous gather (two groups back) must have read it
        frames_ptr, first_id = self.frames_of(i)
        b = self._enq % len(self._out_ptr)
        self._enq += 1
        self.ctx.enqueue(frames_ptr, self.B, first_id, self._out_ptr[b][0], self._out_ptr[b][1], self._rec_ptr[b], self._md_ptr[k][self._pos])
        self._pos += 1
        if self._pos == self.G:       # (gather_every = 0: only when a region outgrows region_steps - the buffer is full)
            self._gather_group()
        self.steps_done = i + 1

    @property
    def out(self):
        """the output buffer the most recent step's records went to"""
        return self.outs[(self._enq - 1) % len(self.outs)]

    @property
    def rec(self):
        return self.recs[(self._enq - 1) % len(self.recs)]

    def flush(self):
        """gather a group the steps so far have left incomplete (gather_every = 0: the region's ONE gather)"""
        if self._pos:
            self._gather_group()

    def fence(self):
        """Barrier + device synchronize on both sides: what brackets a timed region.  Every step's rows have been gathered when it
        returns (the exchange belongs to the region it closes)."""
        self.flush()
        if self.on_gpu:
            self._torch.cuda.synchronize(self.device)
        if self._fence_dist:
            self._fence_dist.barrier()
        if self.on_gpu:
            self._torch.cuda.synchronize(self.device)

    def verify_gather(self):
        """After fence(): the table the LAST gather filled asynchronously must hold EVERY rank's rows of that group at that
        rank's block, on every rank.  Checked against a second, synchronous gather of the same rows; the verdicts are
        combined (min over ranks), so every rank returns the same answer."""
        if not self.dist:
            return None
        if self.last_gathered is None:
            return False
        torch, (k, _) = self._torch, self.last_gathered
        mine, rows = self.md2[k], self.G * self.B
        own = torch.equal(self.md_all2[k][self.rank * rows:(self.rank + 1) * rows], mine)
        blocks = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(blocks, mine)
        whole = torch.equal(self.md_all2[k], torch.cat(blocks, dim=0))
        flag = torch.tensor([1 if (own and whole) else 0], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(int(flag.item()) == 1)
