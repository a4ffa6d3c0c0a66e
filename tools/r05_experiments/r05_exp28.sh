#!/bin/bash
# round 5, experiment 28: k_l2_stats - finds with plain loads, one atomic per streak of lanes with the same root; k_l2_link's finds with plain
# loads as well (the compare-and-swap at the root validates them), same box
O=gpurun_out/r05_exp28.log
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "l2 or level2 or level_2 or dense or all_set or random_config" > $O 2>&1; echo "pytest (l2, product = stats form) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
RC_LIB_PATH=$PWD/ab_build/librecode_hip_l2p.so timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "l2 or level2 or level_2 or dense or all_set or random_config" >> $O 2>&1; echo "pytest (l2, plain link finds) rc=$?" >> $O
A=ab_build/librecode_hip_l2r.so
for cfg in "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--level 2 --sparsity-ppm 10000" "--config 4" "--level 2 --sparsity-ppm 100000"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" old=$A stats=ab_build/librecode_hip_l2s2.so plain=ab_build/librecode_hip_l2p.so >> $O 2>&1 || exit 1
done
echo done >> $O
