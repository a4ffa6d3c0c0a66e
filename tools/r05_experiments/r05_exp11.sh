#!/bin/bash
# round 5, experiment 11: level 2 after the list-overflow fix: the suite; what the directory costs the reduce kernel (a no-directory timing build), alone and pipelined
O=gpurun_out/r05_exp11.log
: > $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "l2" > gpurun_out/r05_exp11_pytest0.log 2>&1; echo "pytest (l2 parity) rc=$?" >> $O; tail -n 3 gpurun_out/r05_exp11_pytest0.log >> $O
if ! grep -q " passed" gpurun_out/r05_exp11_pytest0.log || grep -q "Aborted\|failed" gpurun_out/r05_exp11_pytest0.log; then echo "stopping: level-2 parity tests did not pass" >> $O; exit 1; fi
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp11_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp11_pytest.log >> $O
A=ab_build/librecode_hip
python3 tools/ab_libs.py --rounds 3 --bench "--config 4" main nowb=${A}_nowb.so >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--config 4 --no-pipeline" main nowb=${A}_nowb.so >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --sparsity-ppm 10000" main nowb=${A}_nowb.so >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --sparsity-ppm 10000 --no-pipeline" main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --clustered --sparsity-ppm 2000 --depth 12" main >> $O 2>&1
echo done >> $O
