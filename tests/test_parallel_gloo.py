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
