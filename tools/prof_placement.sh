#!/bin/bash
# per-dispatch counters of the reduce kernel across the allocations of tools/placement_probe.py
set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/placement; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/tools/placement_probe.py > $OUT/run.log 2> $OUT/err.log
cd $REPO
grep trial $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_reduce_tiles" in r["Kernel_Name"]]
by = defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
for t in range(0, len(ids), 24):
    chunk = ids[t:t + 24][4:]
    names = sorted(by[chunk[0]])
    print("trial", t // 24, {n: "%.4g" % (sum(by[i][n] for i in chunk) / len(chunk)) for n in names})
PY
find $OUT -name "*.csv" -size +4M -delete
