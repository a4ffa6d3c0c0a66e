#!/bin/bash
# same-box reduce-kernel timing of the product build and the store-ablation builds for one configuration + SQ counters
# usage: tools/prof_ablate2.sh <tag> <quick_perf args>
TAG=$1; shift
ARGS="$@"
REPO=$(pwd); OUT=$REPO/gpurun_out/abl_$TAG; mkdir -p $OUT
for round in 1 2; do
  for v in main abl1 abl2 abl7; do
    if [ $v = main ]; then unset RC_AB_LIB; else export RC_AB_LIB=$REPO/ab_build/librecode_hip_$v.so; fi
    echo -n "$v: " >> $OUT/timing.log
    python3 tools/quick_perf.py $ARGS 2>&1 | grep shape | sed 's/.*median ms/ms/' >> $OUT/timing.log
  done
done
unset RC_AB_LIB
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tools/prof_pmc.sh ${TAG}_sq "$g" $ARGS 2>&1 | grep reduce >> $OUT/pmc.log
done
cat $OUT/timing.log $OUT/pmc.log
