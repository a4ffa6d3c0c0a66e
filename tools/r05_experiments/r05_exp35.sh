#!/bin/bash
# round 5, experiment 35: phase shares of the reduce kernel on dense frames with the group-by-group compaction
O=gpurun_out/r05_exp35.log
echo "== phase shares (RC_PHASE_TIMING build; s_memtime per phase, lane 0 of one workgroup in 64)" > $O
for a in "4096 4096 32 100000 16 2" "4096 4096 32 300000 16 2" "4096 4096 16 600000 16 2" "4096 4096 32 50000 12 2"; do
  RC_AB_LIB=ab_build/librecode_hip_phase.so timeout -k 10 200 python3 tools/phase_timing.py $a >> $O 2>&1 || exit 1
done
echo done >> $O
