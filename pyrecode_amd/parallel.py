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


def all_gather_rows(rows):
    """rows: int64 ndarray [k, w] (k may differ per rank).  Returns the list of every rank's rows, in rank order.
    One collective for the payload (padded to the largest k) preceded by a tiny one for the counts."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return [rows]
    import torch
    world = dist.get_world_size()
    on_gpu = dist.get_backend() == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
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


def merge_direct(folder_path, base_filename, rank=None, world=None, records=None, index=None):
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
    per_rank = all_gather_rows(rows)
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
    metrics = w.run(data)
    index = w.frame_index()   # what this rank wrote, and where: the merge reads nothing back but the frame data it copies
    w.close()
    if dist is not None and world > 1:
        dist.barrier()  # every part file is complete before anyone reads part 000's header
    nz = None
    if merge:
        base = '%s.rc%d' % (Path(image_filename).stem, int(input_params.reduction_level))
        nz = merge_direct(output_directory, base, rank=rank, world=world, index=index)
    return metrics, nz
