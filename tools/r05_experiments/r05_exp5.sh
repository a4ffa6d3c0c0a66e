#!/bin/bash
O=gpurun_out/r05_exp5.log
: > $O
python3 tools/dbg_allset.py >> $O 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp5_pytest.log 2>&1; echo "pytest rc=$?" >> $O; tail -n 5 gpurun_out/r05_exp5_pytest.log >> $O
A=ab_build/librecode_hip
python3 tools/ab_libs.py --rounds 3 old=${A}_g4.so,RC_OLD_ASSEMBLE=1 g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--clustered --sparsity-ppm 11000 --depth 12" old=${A}_g4.so,RC_OLD_ASSEMBLE=1 g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--config 5" old=${A}_g4.so,RC_OLD_ASSEMBLE=1 g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--sparsity-ppm 100000 --stack 64 --batch 32" g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--sparsity-ppm 300000 --stack 64 --batch 32" g4=${A}_g4.so main >> $O 2>&1
echo done >> $O
