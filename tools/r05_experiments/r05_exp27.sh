#!/bin/bash
# round 5, experiment 27: level 2 with horizontal runs hung in the directory pass (k_l2_link left with the links between rows), same box
O=gpurun_out/r05_exp27.log
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "l2 or level2 or level_2 or dense or all_set or random_config" > $O 2>&1; echo "pytest (l2) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
A=ab_build/librecode_hip_l2d.so
B=ab_build/librecode_hip_l2r.so
for cfg in "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--level 2 --sparsity-ppm 10000" "--config 4" "--level 2 --sparsity-ppm 100000"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" old=$A new=$B >> $O 2>&1 || exit 1
done
tools/prof_bench.sh r05_l2r_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo done >> $O
