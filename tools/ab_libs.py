#!/usr/bin/env python3
"""Same-box A/B of library builds / environments under bench.py, interleaved, with a warm-up first (boxes of this pool drift by several per
cent within a call: read medians over rounds, never single lines).
usage: tools/ab_libs.py [--rounds N] [--bench "<bench.py args>"] name[=lib.so][,ENV=VAL...] ...
   e.g. tools/ab_libs.py --rounds 4 --bench "--config 5" old=ab_build/a.so,RC_OLD_ASSEMBLE=1 new=ab_build/a.so main"""
import argparse
import json
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(variant, bench_args):
    env = dict(os.environ)
    env.pop("RC_LIB_PATH", None)
    name, _, rest = variant.partition("=")
    for tok in (rest.split(",") if rest else []):
        if "=" in tok:
            k, v = tok.split("=", 1)
            env[k] = v
        elif tok:
            env["RC_LIB_PATH"] = os.path.join(REPO, tok)
    cmd = [sys.executable, "bench.py"] + bench_args + ["--steps", "20", "--warmup", "5", "--min-seconds", "0.7", "--no-cpu-baseline", "--no-ingest"]
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    if not lines:
        return None, p.stderr.decode()[-400:]
    return json.loads(lines[-1]), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--bench", default="")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    bench_args = a.bench.split()
    names = [v.partition("=")[0] for v in a.variants]
    print("== bench.py %s | %s" % (a.bench, " ".join(a.variants)), flush=True)
    run(a.variants[0], bench_args)       # warm-up: clocks, page tables, first-import costs
    res = {n: [] for n in names}
    for r in range(a.rounds):
        order = a.variants if r % 2 == 0 else a.variants[::-1]
        for v in order:
            n = v.partition("=")[0]
            j, err = run(v, bench_args)
            if j is None:
                print("  %-14s round %d: ERROR %s" % (n, r, err), flush=True)
                continue
            res[n].append((j["value"], j["roofline"]["kernel_ms"], j["ms_per_step"], j["roofline"]["whole_path_frac"], j["config"]["record_bytes_per_frame"], j["verified"]))
    base = None
    for n in names:
        rows = res[n]
        if not rows:
            continue
        med = statistics.median(x[0] for x in rows)
        base = base or med
        print("  %-14s median %9.0f fps (%+5.1f %%)  kernel %.4f  step %.4f  whole %.3f  rec %.0f  %s   all: %s" % (
            n, med, 100 * (med / base - 1), statistics.median(x[1] for x in rows), statistics.median(x[2] for x in rows),
            statistics.median(x[3] for x in rows), rows[0][4], "ok" if all(x[5] for x in rows) else "NOT VERIFIED",
            " ".join("%.0f" % x[0] for x in rows)), flush=True)


if __name__ == "__main__":
    main()
