#!/bin/bash
# Runs on the MI355X box: the write-stream probe (tools/wr_probe) under rocprofv3, one --pmc pass per counter group
# (TCC has 4 slots per pass), program directly after `--`.  usage: tools/prof_wr_probe.sh <tag>
# Output: gpurun_out/wr_<tag>/{timing.log, table.md}
set -e
TAG=${1:-r02}
REPO=$(pwd); OUT=$REPO/gpurun_out/wr_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$REPO/tools/wr_probe 10 > $OUT/timing.log 2>&1
GROUPS_=(
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"
 "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum"
 "TCC_CYCLE_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum"
 "TCC_WRITE_REQ_sum TCC_READ_REQ_sum TCC_STREAMING_REQ_sum TCC_BUBBLE_sum"
 "TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_IB_REQ_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"
 "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for g in "${GROUPS_[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/g$i -- $REPO/tools/wr_probe 2 > $OUT/g$i.log 2> $OUT/g$i.err || echo "group $i failed: $g" >> $OUT/failed.log
  i=$((i+1))
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict, OrderedDict
out = sys.argv[1]
acc = OrderedDict()
ctrs = []
for f in sorted(glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        c = row["Counter_Name"]
        if c not in ctrs: ctrs.append(c)
        acc.setdefault(k, defaultdict(list))[c].append(float(row["Counter_Value"]))
with open(os.path.join(out, "table.md"), "w") as w:
    ks = [k for k in acc if k.startswith("k_pmc")]
    w.write("| counter | " + " | ".join(ks) + " |\n|---|" + "---|" * len(ks) + "\n")
    for c in ctrs:
        w.write("| %s | " % c + " | ".join("%.4g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "-" for k in ks) + " |\n")
print(open(os.path.join(out, "table.md")).read())
PY
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/timing.log
