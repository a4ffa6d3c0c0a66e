#!/bin/bash
O=gpurun_out/r05_exp2.log
: > $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp2_pytest.log 2>&1; echo "pytest rc=$?" >> $O
tail -n 3 gpurun_out/r05_exp2_pytest.log >> $O
L=ab_build/librecode_hip_knobs.so
tools/r05_gather_grid.sh $L $O
tools/r05_gather_grid.sh $L $O --clustered --sparsity-ppm 11000 --depth 12
tools/r05_gather_grid.sh $L $O --config 5
tools/r05_gather_grid.sh $L $O --config 3
tools/r05_gather_grid.sh ab_build/librecode_hip_knobs_u8.so $O
tools/r05_gather_grid.sh ab_build/librecode_hip_knobs_u8.so $O --config 5
echo done >> $O
