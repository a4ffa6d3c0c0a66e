#!/bin/bash
# End-of-round measurement pass on the MI355X box: tools/final_pass.sh <round-tag>
#   1. unprofiled bench lines of every configuration (+ reader lines + ingest) -> gpurun_out/final_<tag>/
#   2. tools/prof_round.sh passes (kernel-trace stats + FETCH_SIZE / WRITE_SIZE) for the profiled configurations
T=$1
OUT=gpurun_out/final_$T
mkdir -p $OUT
line() { name=$1; shift; python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; tail -c 600 $OUT/$name.json | head -c 400; echo; }
line headline
line cfg3_zstd --scheme 1 --no-cpu-baseline
line zstd_fast --scheme 1 --clevel 0 --no-cpu-baseline --no-ingest
line cfg4 --scheme 8 --level 2 --sparsity-ppm 1000 --no-cpu-baseline --no-ingest
line cfg5_b16 --ny 8184 --nx 11520 --batch 16 --stack 32 --sparsity-ppm 50000 --scheme 1 --depth 12 --no-cpu-baseline --no-ingest
line cfg5_b32 --ny 8184 --nx 11520 --batch 32 --stack 64 --sparsity-ppm 50000 --scheme 1 --depth 12 --no-cpu-baseline --no-ingest
line l3_lz4 --level 3 --no-cpu-baseline --no-ingest
line mode0 --scheme 0 --no-cpu-baseline --no-ingest
line lz4_d12 --depth 12 --no-cpu-baseline --no-ingest
line zstd_d12 --scheme 1 --depth 12 --no-cpu-baseline --no-ingest
line read_zstd --read --scheme 1
line read_lz4 --read --scheme 2
line read_zstd_fast --read --scheme 1 --clevel 0
line read_cfg5 --read --scheme 1 --ny 8184 --nx 11520 --batch 16 --sparsity-ppm 50000 --depth 12
echo "== profiles"
tools/prof_round.sh ${T}_lz4 > $OUT/prof_lz4.log 2>&1
tools/prof_round.sh ${T}_zstd --scheme 1 > $OUT/prof_zstd.log 2>&1
tools/prof_round.sh ${T}_zstd_fast --scheme 1 --clevel 0 > $OUT/prof_zstd_fast.log 2>&1
tools/prof_round.sh ${T}_cfg5_b16 --ny 8184 --nx 11520 --batch 16 --stack 32 --sparsity-ppm 50000 --scheme 1 --depth 12 > $OUT/prof_cfg5_b16.log 2>&1
tools/prof_round.sh ${T}_cfg4 --scheme 8 --level 2 --sparsity-ppm 1000 > $OUT/prof_cfg4.log 2>&1
tools/prof_round.sh ${T}_d12 --depth 12 > $OUT/prof_d12.log 2>&1
echo done
