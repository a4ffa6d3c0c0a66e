#!/bin/bash
# one PMC pass (counters given as one quoted list) over quick_perf: tools/prof_pmc.sh <tag> "<CTR1 CTR2 ...>" <quick_perf args...>
set -e
TAG=$1; CTRS=$2; shift; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pmc -- python3 $REPO/tools/quick_perf.py "$@" > $OUT/run.log 2> $OUT/err.log
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rc::", "")
    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "reduce" in k or "assemble" in k or "fse" in k or "scan" in k:
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
find $OUT -name "*.csv" -size +4M -delete
