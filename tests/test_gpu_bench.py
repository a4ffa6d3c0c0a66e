"""The driver's contract for bench.py (one JSON line, the keys it reads), exercised at a few steps so that a change to the
library cannot silently break the round-end run.  Also through torch.distributed.run with one rank (RCCL path)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_bench_contract(launcher):
    args = ["bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--stack", "64", "--no-cpu-baseline", "--min-seconds", "0.2", "--ingest-frames", "64"]
    if launcher == "plain":
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29533"] + args
    d = _run(cmd)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["unit"] == "frames/s" and d["dtype"] == "u16"
    assert d["value"] > 1000 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "workload" in d["config"] and "4096x4096" in d["config"]["workload"]


@pytest.mark.gpu
def test_bench_cpu_baseline_leg():
    d = _run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--stack", "64", "--min-seconds", "0.2", "--ingest-frames", "64"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no rank in the environment starts two child ranks itself.  On a one-GPU box the product
    form (RCCL) must fail in the CHILDREN with a message that names the cause; the rehearsal form (both ranks on cuda:0, gloo for
    the collective) runs the whole two-rank step loop - double-buffered metadata rows, side-stream gather, every-rank check."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    base = [sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--stack", "64", "--batch", "16", "--min-seconds", "0.2"]
    if torch.cuda.device_count() < 2:
        p = subprocess.run(base, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode != 0 and "needs one GPU per rank" in p.stderr.decode(), p.stderr.decode()[-2000:]
        assert p.stdout.decode().strip() == ""
    p = subprocess.run(base + ["--shared-gpu", "--dist-backend", "gloo"], cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["gather_verified"] is True
    assert d["config"]["collective_ranks"] == 2 and d["config"]["collective_backend"] == "gloo" and d["config"]["shared_gpu_rehearsal"] is True
    assert d["config"]["launch"].startswith("bench.py started its own ranks")
    assert d["cpu_baseline"] is None


@pytest.mark.gpu
def test_bench_fails_when_a_record_fails_the_check():
    """`verified: false` must be an exit code, not only a field: with one byte of a checked record flipped (test switch
    RC_BENCH_CORRUPT_RECORD) the line still appears and says false, and the process returns non-zero.  Without the switch every
    distinct batch of the stack has two records checked - one inside it and its last."""
    args = [sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--stack", "64", "--batch", "16", "--no-cpu-baseline", "--no-ingest",
            "--min-seconds", "0.1"]
    d = _run(args)
    assert d["verified"] is True and len(d["records_checked"]) == 8 and sorted(c["batch"] for c in d["records_checked"]) == [0, 0, 1, 1, 2, 2, 3, 3]
    assert sum(1 for c in d["records_checked"] if c["record"] == 15) == 4          # every batch's last record is among them
    assert all(c["ok"] for c in d["records_checked"])
    assert d["host_enqueue_us_per_step"] > 0 and d["roofline"]["pattern_floor_ms"]["reads_only"] > 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RC_BENCH_CORRUPT_RECORD="1")
    p = subprocess.run(args, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0, "a corrupted record went unnoticed"
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["verified"] is False
    assert "verified: false" in p.stderr.decode()


@pytest.mark.gpu
@pytest.mark.parametrize("gather_every", [0, 3])
def test_bench_gather_every(gather_every):
    """Two ranks on the one GPU (gloo rehearsal): the metadata gather once per timed region (0) and every third step."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "7", "--warmup", "2", "--stack", "32", "--batch", "16", "--min-seconds", "0.1",
           "--shared-gpu", "--dist-backend", "gloo", "--gather-every", str(gather_every)]
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["gather_verified"] is True and d["config"]["gather_every"] == gather_every


@pytest.mark.gpu
@pytest.mark.parametrize("gather_every", [0, 1])
def test_bench_under_torchrun_with_a_one_rank_rccl_communicator(gather_every):
    """What the driver's N > 1 run does, with the one GPU a test box has: torch.distributed.run, backend nccl (= RCCL), the metadata
    all-gather issued on the device - once per timed region (0, the default) and every step (1).  A real all_gather_into_tensor runs,
    its table is verified, and every rank says on stderr where it runs (the first lines of a scaling log)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29541 + gather_every), "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--stack", "64",
           "--no-cpu-baseline", "--no-ingest", "--min-seconds", "0.2", "--dist-backend", "nccl", "--gather-every", str(gather_every)]
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["gather_verified"] is True and d["verified"] is True
    assert d["config"]["gather_every"] == gather_every
    start = [l for l in p.stderr.decode().splitlines() if l.startswith("bench.py start:")]
    assert len(start) == 1 and "rank 0 of world 1" in start[0] and "visible GPUs" in start[0] and "cuda:0" in start[0] and "backend nccl" in start[0]
