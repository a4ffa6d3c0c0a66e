#!/bin/bash
# kernel-trace (timestamps) of a short pipelined bench run, kept for offline timeline analysis
set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log
cd $REPO
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "synth" not in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-40:]:
    n = r["Kernel_Name"].split("(")[0].replace("void rc::", "")[:28]
    print("%-28s q=%s start %9.1f us  dur %7.1f us" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
