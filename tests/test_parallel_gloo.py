"""CPU, world_size 2 over gloo: the multi-GPU driver's host logic - ownership rule, the metadata all-gather and the direct
merged-file write - must reproduce the reference's merged file from the reference's own part files."""
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import GOLDEN, REPO

FILES = os.path.join(GOLDEN, "files")


def test_frame_block_rule():
    from pyrecode_amd.parallel import frame_block
    assert [frame_block(8, 3, r) for r in range(3)] == [(0, 3), (3, 3), (6, 2)]
    assert [frame_block(9, 3, r) for r in range(3)] == [(0, 3), (3, 3), (6, 3)]
    assert [frame_block(2, 4, r) for r in range(4)] == [(0, 1), (1, 1), (2, 0), (3, 0)]
    assert frame_block(8192, 8, 7) == (7168, 1024)


def test_single_process_merge_direct(tmp_path):
    # world 1 cannot cover a multi-part fixture, so merge the parts pairwise through the same code path rank by rank
    from pyrecode_amd import parallel
    base = "g3_l1z16.rc1"
    for i in range(2):
        shutil.copy(os.path.join(FILES, "%s_part%03d" % (base, i)), tmp_path)
    recs = []
    for i in range(2):
        _, r = parallel.read_part_records(os.path.join(tmp_path, "%s_part%03d" % (base, i)))
        recs += r
    n = parallel.merge_direct(str(tmp_path), base, rank=0, world=1, records=recs)
    assert n == 5
    assert (tmp_path / base).read_bytes() == open(os.path.join(FILES, base), "rb").read()


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(repo)r)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    from pyrecode_amd import parallel
    n = parallel.merge_direct(%(folder)r, %(base)r)
    assert n == %(nz)d, n
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("base,nz", [("g3_l1z16.rc1", 5), ("g3_l3z.rc3", 4), ("g3_l1ro16.rc1", 4)])
def test_two_rank_direct_merge_matches_reference(base, nz, tmp_path):
    for i in range(2):
        shutil.copy(os.path.join(FILES, "%s_part%03d" % (base, i)), tmp_path)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(repo=REPO, folder=str(tmp_path), base=base, nz=nz))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    assert (tmp_path / base).read_bytes() == open(os.path.join(FILES, base), "rb").read()


# ---- cfg-3 proportions without a GPU: fabricated part files (reference layout, arbitrary payload bytes) -------------------
def _fabricate_parts(tmp_path, base, n_frames, world, seed):
    """Part files in the reference's layout for n_frames frames split by the contiguous-block rule: header copied from a
    reference-written fixture (nz patched), records `u32 frame_id | u32 cb | u32 cp | u32 npk | blob`.  Returns the merged
    file's expected bytes, assembled independently of the code under test."""
    import struct
    from pyrecode_amd.parallel import frame_block
    from pyrecode_amd.recode_header import ReCoDeHeader
    rng = np.random.default_rng(seed)
    head = open(os.path.join(FILES, "g3_l1z16.rc1_part000"), "rb").read(512)   # L1, mode 1: three metadata fields
    h = ReCoDeHeader()
    nz_pos, nz_len = h.get_field_position_in_bytes("nz") if hasattr(h, "get_field_position_in_bytes") else 23, 4
    table, blobs = [], []
    for r in range(world):
        lo, cnt = frame_block(n_frames, world, r)
        with open(tmp_path / ("%s_part%03d" % (base, r)), "wb") as f:
            f.write(head[:nz_pos] + int(cnt).to_bytes(nz_len, "little") + head[nz_pos + nz_len:])
            for fid in range(lo, lo + cnt):
                cb, cp = int(rng.integers(1, 4000)), int(rng.integers(0, 3000))
                blob = rng.integers(0, 256, cb + cp, dtype=np.uint8).tobytes()
                f.write(struct.pack("<IIII", fid, cb, cp, cp + 7) + blob)
                table.append((cb, cp, cp + 7))
                blobs.append(blob)
    import struct as _s
    return (head[:nz_pos] + int(n_frames).to_bytes(nz_len, "little") + head[nz_pos + nz_len:] +
            b"".join(_s.pack("<III", *row) for row in table) + b"".join(blobs))


@pytest.mark.parametrize("n_frames", [150, 1])   # 75 frames per rank; one frame: rank 1's part file is empty
def test_two_rank_direct_merge_at_scale_and_with_an_empty_rank(n_frames, tmp_path):
    from pyrecode_amd.recode_reader import merge_parts
    base = "fab.rc1"
    want = _fabricate_parts(tmp_path, base, n_frames, 2, 11 + n_frames)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(repo=REPO, folder=str(tmp_path), base=base, nz=n_frames))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31500 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert (tmp_path / base).read_bytes() == want
    os.remove(tmp_path / base)
    merge_parts(str(tmp_path), base, 2)          # the file-based merge (streaming, two passes) must agree
    assert (tmp_path / base).read_bytes() == want


# ---- the bench's step loop (pyrecode_amd.parallel.ShardedStepLoop) with two ranks: device calls stubbed at the ReduceContext
# boundary, so the double-buffered metadata rows, the side-"stream" gather and the every-rank check run with world = 2 ----
LOOP_WORKER = textwrap.dedent("""
    import ctypes, os, sys
    import numpy as np
    sys.path.insert(0, %(repo)r)
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrecode_amd.parallel import ShardedStepLoop

    B = 5

    class StubCtx:                      # what _lib.ReduceContext offers the loop; rows are a function of (rank, step)
        device_id = None
        waits = 0

        def enqueue(self, frames_ptr, n, first_id, out_ptr, out_cap, rec_ptr, md_ptr):
            md = np.ctypeslib.as_array((ctypes.c_int32 * (n * 3)).from_address(md_ptr)).reshape(n, 3)
            md[:, 0] = first_id + np.arange(n)
            md[:, 1] = frames_ptr
            md[:, 2] = rank + 1

        def wait_results(self, stream_handle):
            self.waits += 1

    def rows(r, i):
        t = np.zeros((B, 3), np.int32)
        t[:, 0] = r * 1000 + i * B + np.arange(B)
        t[:, 1] = 7 * i + r
        t[:, 2] = r + 1
        return t

    out, rec = torch.zeros(64, dtype=torch.uint8), torch.zeros(B + 1, dtype=torch.int64)
    nsteps = 7

    def group_rows(r, steps, G):            # a rank's block of a gathered table: the group's steps, then what an earlier group left there
        t = np.zeros((G * B, 3), np.int32)
        for j, i in enumerate(steps):
            t[j * B:(j + 1) * B] = rows(r, i)
        return t

    # gather_every = 1: one collective per step, double-buffered (the form of rounds 1-3)
    ctx = StubCtx()
    loop = ShardedStepLoop(ctx, B, lambda i: (7 * i + rank, rank * 1000 + i * B), out, rec, torch.device("cpu"))
    assert loop.world == world and loop.rank == rank
    loop.fence()
    for i in range(nsteps):
        loop.step(i)
    loop.fence()
    assert ctx.waits == nsteps and loop.gathers_issued == nsteps
    for i in (nsteps - 1, nsteps - 2):       # both buffers: the last step's table and the one before it
        want = np.concatenate([rows(r, i) for r in range(world)])
        assert np.array_equal(loop.md_all2[i & 1].numpy(), want), (rank, i)
    assert loop.verify_gather() is True
    if rank == world - 1:                    # one rank's table damaged: EVERY rank must learn it
        loop.md_all2[(nsteps - 1) & 1][0, 0] += 1
    assert loop.verify_gather() is False

    # gather_every = 3: groups of three steps; the fence gathers the incomplete last group (7 = 3 + 3 + 1)
    ctx = StubCtx()
    loop = ShardedStepLoop(ctx, B, lambda i: (7 * i + rank, rank * 1000 + i * B), out, rec, torch.device("cpu"), gather_every=3)
    loop.fence()
    for i in range(nsteps):
        loop.step(i)
        assert loop.gathers_issued == (i + 1) // 3
    loop.fence()
    assert loop.gathers_issued == 3 and ctx.waits == 3
    want1 = np.concatenate([group_rows(r, [3, 4, 5], 3) for r in range(world)])           # group 1 -> buffer 1
    assert np.array_equal(loop.md_all2[1].numpy(), want1)
    want2 = np.concatenate([np.concatenate([rows(r, 6), rows(r, 1), rows(r, 2)]) for r in range(world)])   # group 2 (step 6) over group 0's rows
    assert np.array_equal(loop.md_all2[0].numpy(), want2)
    assert loop.verify_gather() is True

    # gather_every = 0: ONE gather per fenced region (BASELINE: "RCCL only for the final merged-index gather")
    ctx = StubCtx()
    loop = ShardedStepLoop(ctx, B, lambda i: (7 * i + rank, rank * 1000 + i * B), out, rec, torch.device("cpu"), gather_every=0, region_steps=nsteps)
    loop.fence()
    assert loop.gathers_issued == 0
    for region, n_here in enumerate((nsteps, nsteps - 2)):      # a full region (its last step issues the gather), a shorter one (the fence does)
        first = region * nsteps
        for i in range(n_here):
            loop.step(first + i)
            assert loop.gathers_issued == region + (1 if i + 1 == nsteps else 0)
        loop.fence()
        assert loop.gathers_issued == region + 1 and ctx.waits == region + 1
        steps = list(range(first, first + n_here))
        want = np.concatenate([group_rows(r, steps, nsteps) for r in range(world)])
        assert np.array_equal(loop.md_all2[region & 1].numpy()[:, :], want) if n_here == nsteps else \
            all(np.array_equal(loop.md_all2[region & 1].numpy()[r * nsteps * B:r * nsteps * B + n_here * B], group_rows(r, steps, n_here)) for r in range(world))
    assert loop.verify_gather() is True
    if rank == 0:
        loop.md_all2[1][3, 1] ^= 1
    assert loop.verify_gather() is False
    dist.destroy_process_group()
""")


def _run_ranks(script_path, world=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() * 7 + 3 + world) % 2000), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script_path)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r])


@pytest.mark.parametrize("world", [2, 8])
def test_step_loop_over_gloo(tmp_path, world):
    """the bench's step loop with 2 ranks, and with the 8 the scaling run uses (one process per GPU of a node)"""
    script = tmp_path / "loop_worker.py"
    script.write_text(LOOP_WORKER % dict(repo=REPO))
    _run_ranks(script, world)


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no rank in the environment must launch two CHILD ranks (never touching the GPU itself);
    here there is no GPU, so both children refuse with a message of their own and the launcher's exit code says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode()
    assert p.returncode != 0
    assert "needs one GPU per rank" in err, err[-2000:]
    assert "2-rank child job failed" in err
    assert p.stdout.decode().strip() == ""
