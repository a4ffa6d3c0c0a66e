#!/bin/bash
# round 5, experiment 15: level-2 reduce kernel with three-wave workgroups (development knob); level-2 kernel trace at 1 %
O=gpurun_out/r05_exp15.log
: > $O
A=ab_build/librecode_hip_knobs.so
python3 tools/ab_libs.py --rounds 3 --bench "--level 2 --sparsity-ppm 10000" rw4=$A rw3=$A,RC_REDUCE_WG_WAVES=3 >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--config 4" rw4=$A rw3=$A,RC_REDUCE_WG_WAVES=3 >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --clustered --sparsity-ppm 2000 --depth 12" rw4=$A rw3=$A,RC_REDUCE_WG_WAVES=3 >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--config 3" rw4=$A rw3=$A,RC_REDUCE_WG_WAVES=3 >> $O 2>&1
tools/prof_bench.sh r05_l2v6_1pct --level 2 --sparsity-ppm 10000 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2v6_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo done >> $O
