#!/bin/bash
# round 5, experiment 25: level 2 in its small-LDS form with two chains (the product default): the suite, one chain against two on the same
# box, and against round 4's tree
O=gpurun_out/r05_exp25.log
: > $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_exp25_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp25_pytest.log >> $O
grep -q "rc=0" $O || exit 1
D=ab_build/librecode_hip_l2d.so
for cfg in "--config 4" "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--level 2 --sparsity-ppm 100000"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" one=$D,RC_ONE_CHAIN=1 two=$D >> $O 2>&1 || exit 1
done
tools/ab_trees.sh ab_build/r04_tree 3 >> $O 2>&1 <<CFGS
--level 2 --sparsity-ppm 10000
--level 2 --clustered --sparsity-ppm 2000 --depth 12
--config 4
CFGS
echo done >> $O
