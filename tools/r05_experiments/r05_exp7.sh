#!/bin/bash
# round 5, experiment 7: where dense frames lose their time (phase shares at 1 / 10 / 30 %), batch size, gaps between reduce kernels
O=gpurun_out/r05_exp7.log
: > $O
echo "== phase shares (RC_PHASE_TIMING build; s_memtime per phase, lane 0 of one workgroup in 64)" >> $O
for a in "4096 4096 32 10000 16 2" "4096 4096 32 100000 16 2" "4096 4096 32 300000 16 2" "4096 4096 32 300000 16 0"; do
  RC_AB_LIB=ab_build/librecode_hip_phase.so python3 tools/phase_timing.py $a >> $O 2>&1
done
echo "== batch size (frames per call)" >> $O
python3 tools/ab_libs.py --rounds 3 b64=,X=1 >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--batch 128 --stack 256" b128=,X=1 >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--batch 256 --stack 256" b256=,X=1 >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--batch 32 --stack 256" b32=,X=1 >> $O 2>&1
echo "== kernel trace with timestamps: gaps between consecutive reduce kernels" >> $O
REPO=$(pwd); OUT=$REPO/gpurun_out/kt_r05_gaps; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --steps 20 --warmup 5 --min-seconds 0.3 --no-cpu-baseline --no-ingest > $OUT/line.json 2> $OUT/kt.err)
python3 - >> $O 2>&1 <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt_r05_gaps/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
red = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_reduce_tiles" in r["Kernel_Name"]])
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(red, red[1:])]
durs = [(e - s) / 1e3 for s, e in red]
import statistics as st
g = [x for x in gaps if x < 200]
print("reduce kernels %d: duration median %.1f us; gap end->next start median %.1f us (p10 %.1f, p90 %.1f); overlapping starts %d" % (len(red), st.median(durs), st.median(g), sorted(g)[len(g)//10], sorted(g)[9*len(g)//10], sum(1 for x in gaps if x < 0)))
for name in ("k_gather", "k_scan_frames", "k_layout"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    if d: print("%s: n %d median %.1f us" % (name, len(d), st.median(d)))
PY
rm -rf $OUT/kt
echo done >> $O
