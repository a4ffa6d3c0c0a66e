#!/bin/bash
# round 5, experiment 34: k_gather's items - tiles per item halved until a batch has 1024 (product) / 2048 / 4096 / 8192 items, same box
O=gpurun_out/r05_exp34.log
: > $O
D=ab_build/librecode_hip_gmi.so
for cfg in "--sparsity-ppm 300000 --stack 64 --batch 32" "--sparsity-ppm 600000 --stack 32 --batch 16" "--sparsity-ppm 100000 --stack 64 --batch 32" "--batch 32" "" "--clustered --sparsity-ppm 11000 --depth 12"; do
  python3 tools/ab_libs.py --rounds 2 --bench "$cfg" m1024=$D m2048=$D,RC_GATHER_MIN_ITEMS=2048 m4096=$D,RC_GATHER_MIN_ITEMS=4096 m8192=$D,RC_GATHER_MIN_ITEMS=8192 m16k=$D,RC_GATHER_MIN_ITEMS=16384 >> $O 2>&1 || exit 1
done
echo done >> $O
