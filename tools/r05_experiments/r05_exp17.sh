#!/bin/bash
# round 5, experiment 17: level-2 statistics over the list of hung pixels; configs[0]; the rest of the profiles
O=gpurun_out/r05_exp17.log
: > $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "l2 or random_config" > gpurun_out/r05_exp17_pytest0.log 2>&1; echo "pytest (l2) rc=$?" >> $O; tail -n 3 gpurun_out/r05_exp17_pytest0.log >> $O
if ! grep -q " passed" gpurun_out/r05_exp17_pytest0.log || grep -q "Aborted\|failed" gpurun_out/r05_exp17_pytest0.log; then echo "stopping: level-2 tests did not pass" >> $O; exit 1; fi
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp17_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp17_pytest.log >> $O
tools/ab_trees.sh ab_build/r04_tree 3 >> $O 2>&1 <<CFGS
--level 2 --sparsity-ppm 10000
--level 2 --clustered --sparsity-ppm 2000 --depth 12
--config 4
--config 1
CFGS
tools/prof_bench.sh r05_l2v7_1pct --level 2 --sparsity-ppm 10000 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/final_profiles.sh r05a b >> $O 2>&1
echo done >> $O
