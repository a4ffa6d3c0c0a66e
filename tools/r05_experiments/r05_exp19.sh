#!/bin/bash
# round 5, experiment 19: phase shares of the reduce kernel where the whole path is furthest from the roofline (cfg 5, zstd, level 2)
O=gpurun_out/r05_exp19.log
echo "== phase shares (RC_PHASE_TIMING build; s_memtime per phase, lane 0 of one workgroup in 64)" > $O
for a in "8184 11520 16 50000 12 1" "4096 4096 32 10000 16 1" "4096 4096 32 50000 12 1" "4096 4096 32 50000 12 2" "4096 4096 32 10000 16 2 2" "4096 4096 32 1000 16 8 2"; do
  RC_AB_LIB=ab_build/librecode_hip_phase.so timeout -k 10 200 python3 tools/phase_timing.py $a >> $O 2>&1 || exit 1
done
echo done >> $O
