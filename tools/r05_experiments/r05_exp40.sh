#!/bin/bash
# round 5, experiment 40: k_l2_dir without the per-tile load of the previous tile's last word (carried through a readlane), same box
O=gpurun_out/r05_exp40.log
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "l2 or level2 or level_2 or random_config" > $O 2>&1; echo "pytest (l2) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
for cfg in "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--level 2 --sparsity-ppm 10000" "--config 4"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" old=ab_build/librecode_hip_dirold.so new=ab_build/librecode_hip_dirnew.so >> $O 2>&1 || exit 1
done
echo done >> $O
