#!/bin/bash
# kernel-trace stats of one quick_perf configuration on the MI355X box: tools/prof_kt.sh <tag> <quick_perf args...>
set -e
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/tools/quick_perf.py "$@" > $OUT/run.log 2> $OUT/kt.err
cd $REPO
python3 tools/summarize_rocprof.py $OUT > $OUT/summary.md
find $OUT -name "*_kernel_trace.csv" -delete
cat $OUT/run.log | grep shape; cat $OUT/summary.md
