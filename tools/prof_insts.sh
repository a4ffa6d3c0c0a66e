#!/bin/bash
# SQ instruction counts of the reduce kernel (every instantiation launched) for a few configurations: tools/prof_insts.sh [cfg...]
# quick_perf arguments: ny nx B ppm depth scheme [level]
REPO=$(pwd)
if [ $# -eq 0 ]; then set -- "4096 4096 64 10000 16 2" "4096 4096 64 10000 16 1" "4096 4096 64 10000 16 0" "4096 4096 64 10000 16 2 3" "8184 11520 16 50000 12 1" "8184 11520 16 50000 12 0" "8184 11520 16 50000 12 1 3" "4096 4096 32 1000 12 8 2"; fi
for cfg in "$@"; do
  tag=insts_$(echo $cfg | tr ' ' '_')
  rm -rf $REPO/gpurun_out/pmc_$tag
  tools/prof_pmc.sh $tag "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $cfg 2>&1 | grep "k_reduce_tiles<4, true, true" | grep -v "SQ_WAVES': '4096'\|SQ_WAVES': '2.304e+04'" | sed "s/^/$cfg :: /" | sed "s/'SQ_BUSY_CYCLES': '[0-9.e+]*', //"
done
