#!/bin/bash
# round 5, experiment 33: dense tiles compacted group by group (one 16-byte read per lane and group, three packed scans, predicated writes)
# instead of window by window through the small stage; same box
O=gpurun_out/r05_exp33.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "dense or all_set or fuzz or random or l2 or level2 or fullsize" > $O 2>&1; echo "pytest (dense, fuzz, l2, fullsize) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
A=ab_build/librecode_hip_l2s3.so
B=ab_build/librecode_hip_dense.so
for cfg in "--sparsity-ppm 100000 --stack 64 --batch 32" "--sparsity-ppm 300000 --stack 64 --batch 32" "--sparsity-ppm 600000 --stack 32 --batch 16" "--sparsity-ppm 100000 --stack 64 --batch 32 --scheme 1 --depth 12" "--level 2 --sparsity-ppm 100000 --stack 64 --batch 32" "--config 5" ""; do
  python3 tools/ab_libs.py --rounds 2 --bench "$cfg" old=$A new=$B >> $O 2>&1 || exit 1
done
echo done >> $O
