#!/bin/bash
# round 5, experiment 36: the group-by-group compaction without branches (unset pixels write to the lane's entry of the unused small stage)
O=gpurun_out/r05_exp36.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "dense or all_set or fuzz or random or l2 or level2 or fullsize" > $O 2>&1; echo "pytest (dense, fuzz, l2, fullsize) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
for a in "4096 4096 32 100000 16 2" "4096 4096 32 300000 16 2"; do
  RC_AB_LIB=ab_build/librecode_hip_phase.so timeout -k 10 200 python3 tools/phase_timing.py $a 2>&1 | grep -v amdgpu.ids >> $O || exit 1
done
A=ab_build/librecode_hip_dense.so
B=ab_build/librecode_hip_dense2.so
for cfg in "--sparsity-ppm 100000 --stack 64 --batch 32" "--sparsity-ppm 300000 --stack 64 --batch 32" "--sparsity-ppm 600000 --stack 32 --batch 16" "--sparsity-ppm 100000 --stack 64 --batch 32 --scheme 1 --depth 12" ""; do
  python3 tools/ab_libs.py --rounds 2 --bench "$cfg" branches=$A nobranch=$B >> $O 2>&1 || exit 1
done
echo done >> $O
