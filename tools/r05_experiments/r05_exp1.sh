#!/bin/bash
# round 5, experiment 1: what the second stage costs the pipelined step (skip builds), one-wave second-stage workgroups
O=gpurun_out/r05_exp1.log
: > $O
B="--steps 20 --warmup 5 --min-seconds 0.7"
echo "== headline skip bits (0 none, 2 scans, 16 assemble, 18 scans+assemble, 26 scans+layout+assemble)" >> $O
tools/skip_bench.sh "$B" 0 2 16 18 26 >> $O 2>&1
echo "== detector lz4 skip" >> $O
tools/skip_bench.sh "$B --clustered --sparsity-ppm 11000 --depth 12" 0 16 26 >> $O 2>&1
echo "== cfg5 skip (1 fse n/a, 16 assemble, 26)" >> $O
tools/skip_bench.sh "$B --config 5" 0 16 26 >> $O 2>&1
unset RC_LIB_PATH RC_DEV_SKIP_BITS
echo "== A/B small wg: headline" >> $O
tools/ab_bench.sh "$B" main ab_build/librecode_hip_smallwg.so ab_build/librecode_hip_scan64.so >> $O 2>&1
echo "== A/B small wg: detector lz4" >> $O
tools/ab_bench.sh "$B --clustered --sparsity-ppm 11000 --depth 12" main ab_build/librecode_hip_smallwg.so >> $O 2>&1
echo "== A/B small wg: cfg5" >> $O
tools/ab_bench.sh "$B --config 5" main ab_build/librecode_hip_smallwg.so >> $O 2>&1
echo "== A/B small wg: cfg3" >> $O
tools/ab_bench.sh "$B --config 3" main ab_build/librecode_hip_smallwg.so >> $O 2>&1
echo "== decompose headline" >> $O
tools/decompose.sh >> $O 2>&1
echo done >> $O
