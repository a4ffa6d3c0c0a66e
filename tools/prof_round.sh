#!/bin/bash
# Runs on the MI355X box (via gpurun): kernel-trace stats + the two PMC passes of bench.py for one configuration.
# usage: tools/prof_round.sh <tag> [bench.py arguments ...]      outputs under gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --steps 20 --warmup 3 --min-seconds 0.1 --no-cpu-baseline --no-ingest "$@" > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --steps 5 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-ingest "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $REPO/bench.py --steps 5 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-ingest "$@" > /dev/null 2> $OUT/write.err
cd $REPO
python3 tools/summarize_rocprof.py $OUT > $OUT/summary.md
# keep only the small files (gpurun_out merge limit)
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/summary.md
