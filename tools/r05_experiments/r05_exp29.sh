#!/bin/bash
# round 5, experiment 29: k_l2_stats joins streaks only where more than eight lanes of the 64 are hung (product), same box
O=gpurun_out/r05_exp29.log
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "l2 or level2 or level_2 or dense or all_set or random_config" > $O 2>&1; echo "pytest (l2) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
for cfg in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--level 2 --sparsity-ppm 100000"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" old=ab_build/librecode_hip_l2r.so always=ab_build/librecode_hip_l2s2.so over8=ab_build/librecode_hip_l2s3.so >> $O 2>&1 || exit 1
done
echo done >> $O
