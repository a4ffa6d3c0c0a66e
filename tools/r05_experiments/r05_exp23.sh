#!/bin/bash
# round 5, experiment 23: two chains - consecutive batches' second stages on two streams (RC_TWO_CHAINS, development build), same box
O=gpurun_out/r05_exp23.log
: > $O
D=ab_build/librecode_hip_l2c.so
for cfg in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--config 4" "" "--scheme 1" "--config 5"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" one=$D two=$D,RC_TWO_CHAINS=1 >> $O 2>&1 || exit 1
done
echo done >> $O
