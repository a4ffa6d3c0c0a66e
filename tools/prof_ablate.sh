#!/bin/bash
# Runs on the MI355X box: (1) same-box timing of the reduce kernel for the product build and the store-ablation builds
# (tools/build_ablate.sh), three interleaved rounds; (2) SQ / TCC counter passes of the product build and of the
# all-stores-dropped build.  usage: tools/prof_ablate.sh <tag> [quick_perf args]      -> gpurun_out/abl_<tag>/
TAG=${1:-r02}; shift
ARGS=${@:-4096 4096 64 10000 16 2}
REPO=$(pwd); OUT=$REPO/gpurun_out/abl_$TAG; mkdir -p $OUT
for round in 1 2 3; do
  for v in main abl1 abl2 abl4 abl7; do
    if [ $v = main ]; then unset RC_AB_LIB; else export RC_AB_LIB=$REPO/ab_build/librecode_hip_$v.so; fi
    echo -n "$v: " >> $OUT/timing.log
    python3 tools/quick_perf.py $ARGS 2>&1 | grep shape >> $OUT/timing.log
  done
done
unset RC_AB_LIB
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
         "TCC_CYCLE_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  for v in main abl7; do
    if [ $v = main ]; then unset RC_AB_LIB; else export RC_AB_LIB=$REPO/ab_build/librecode_hip_$v.so; fi
    echo "== $v group $i" >> $OUT/pmc.log
    tools/prof_pmc.sh ${TAG}_${v}_$i "$g" $ARGS 2>&1 | grep reduce >> $OUT/pmc.log
  done
  i=$((i+1))
done
cat $OUT/timing.log; cat $OUT/pmc.log
