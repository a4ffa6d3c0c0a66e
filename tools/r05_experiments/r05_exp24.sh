#!/bin/bash
# round 5, experiment 24: two chains with the second chain's stream at another priority (a hardware queue of its own?), same box
O=gpurun_out/r05_exp24.log
: > $O
D=ab_build/librecode_hip_l2c.so
for cfg in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" ""; do
  GPU_MAX_HW_QUEUES=8 python3 tools/ab_libs.py --rounds 2 --bench "$cfg" one=$D two=$D,RC_TWO_CHAINS=1 low=$D,RC_TWO_CHAINS=2 high=$D,RC_TWO_CHAINS=3 >> $O 2>&1 || exit 1
done
echo "== default GPU_MAX_HW_QUEUES" >> $O
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --sparsity-ppm 10000" one=$D low=$D,RC_TWO_CHAINS=2 high=$D,RC_TWO_CHAINS=3 >> $O 2>&1
echo done >> $O
