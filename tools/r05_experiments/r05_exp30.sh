#!/bin/bash
# round 5, experiment 30: the reduce kernel's packet carries its own events (no marker packets on its stream); any-order launch, same box
O=gpurun_out/r05_exp30.log
: > $O
D=ab_build/librecode_hip_ke.so
for cfg in "" "--scheme 1" "--config 5" "--level 2 --sparsity-ppm 10000" "--batch 32"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" base=$D ke1=$D,RC_KERNEL_EVENTS=1 ke2=$D,RC_KERNEL_EVENTS=2 >> $O 2>&1 || exit 1
done
echo done >> $O
