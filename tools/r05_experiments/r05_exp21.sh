#!/bin/bash
# round 5, experiment 21: level 2 within 2.6 KB of LDS and 64 registers per workgroup (directory in global memory, lists of 256 / 1024) against
# the form before it (16.6 / 12.8 KB), same box
O=gpurun_out/r05_exp21.log
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "l2 or level2 or level_2 or dense or all_set" > $O 2>&1; echo "pytest (l2) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
A=ab_build/librecode_hip_l2old.so
B=ab_build/librecode_hip_l2s.so
for cfg in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--config 4" "--level 2 --sparsity-ppm 100000"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" old=$A new=$B >> $O 2>&1 || exit 1
done
echo done >> $O
