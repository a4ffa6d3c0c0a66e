#!/bin/bash
# round 5, experiment 22: the level-2 chain kernel by kernel in its small-LDS form
O=gpurun_out/r05_exp22.log
: > $O
tools/prof_bench.sh r05_l2s_1pct --level 2 --sparsity-ppm 10000 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2s_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo done >> $O
