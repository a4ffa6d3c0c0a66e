#!/bin/bash
# round 5, experiment 14: level 2 with the word bases in the link kernel's own LDS (the reduce kernel as in round 4), small tiles finished by their lane;
# round 4's tree against this one on the level-2 and the BASELINE configurations, same box
O=gpurun_out/r05_exp14.log
: > $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "l2 or random_config" > gpurun_out/r05_exp14_pytest0.log 2>&1; echo "pytest (l2) rc=$?" >> $O; tail -n 3 gpurun_out/r05_exp14_pytest0.log >> $O
if ! grep -q " passed" gpurun_out/r05_exp14_pytest0.log || grep -q "Aborted\|failed" gpurun_out/r05_exp14_pytest0.log; then echo "stopping: level-2 tests did not pass" >> $O; exit 1; fi
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp14_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp14_pytest.log >> $O
tools/ab_trees.sh ab_build/r04_tree 3 >> $O 2>&1 <<CFGS
--config 4
--level 2 --sparsity-ppm 10000
--level 2 --clustered --sparsity-ppm 2000 --depth 12
--config 2
--scheme 0
--config 3
CFGS
tools/prof_bench.sh r05_l2v6_cfg4 --config 4 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo done >> $O
