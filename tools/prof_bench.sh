#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py command on the MI355X box: tools/prof_bench.sh <tag> <bench.py args...>
# Leaves gpurun_out/kt_<tag>/{summary.md, line.json}; copy what should be judged into profiles/.
set -e
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py "$@" > $OUT/line.json 2> $OUT/kt.err
cd $REPO
python3 tools/summarize_rocprof.py $OUT > $OUT/summary.md
find $OUT -name "*_kernel_trace.csv" -delete
cat $OUT/summary.md
