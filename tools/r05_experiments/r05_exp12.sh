#!/bin/bash
# round 5, experiment 12: level-2 link over the listed non-empty words; kernel events around every 4th launch
O=gpurun_out/r05_exp12.log
: > $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "l2 or random_config" > gpurun_out/r05_exp12_pytest0.log 2>&1; echo "pytest (l2) rc=$?" >> $O; tail -n 3 gpurun_out/r05_exp12_pytest0.log >> $O
if ! grep -q " passed" gpurun_out/r05_exp12_pytest0.log || grep -q "Aborted\|failed" gpurun_out/r05_exp12_pytest0.log; then echo "stopping: level-2 tests did not pass" >> $O; exit 1; fi
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp12_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp12_pytest.log >> $O
python3 tools/ab_libs.py --rounds 3 --bench "--config 4" main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--config 4 --no-pipeline" main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --sparsity-ppm 10000" main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --sparsity-ppm 10000 --no-pipeline" main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --clustered --sparsity-ppm 2000 --depth 12" main >> $O 2>&1
echo "== kernel events around every launch / every 4th" >> $O
python3 tools/ab_libs.py --rounds 4 every1=,X=1 >> $O 2>&1
python3 tools/ab_libs.py --rounds 4 --bench "--kernel-events-every 4" every4=,X=1 >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 every1=,X=1 >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--kernel-events-every 4" every4=,X=1 >> $O 2>&1
tools/prof_bench.sh r05_l2v5_cfg4_alone --config 4 --no-pipeline --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo done >> $O
