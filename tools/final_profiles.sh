#!/bin/bash
# End-of-round profiles on the MI355X box: tools/final_profiles.sh <round-tag> [which]   (the second half of tools/final_pass.sh, callable by itself;
# which = "a" (default: the BASELINE configurations) or "b" (the others))
T=$1; W=${2:-a}
OUT=gpurun_out/final_$T
mkdir -p $OUT
if [ "$W" = "a" ]; then
tools/prof_round.sh ${T}_lz4 > $OUT/prof_lz4.log 2>&1
tools/prof_round.sh ${T}_zstd --scheme 1 > $OUT/prof_zstd.log 2>&1
tools/prof_round.sh ${T}_cfg5 --config 5 > $OUT/prof_cfg5.log 2>&1
tools/prof_round.sh ${T}_cfg4 --config 4 > $OUT/prof_cfg4.log 2>&1
tools/prof_round.sh ${T}_d12 --depth 12 > $OUT/prof_d12.log 2>&1
tools/prof_round.sh ${T}_det_lz4 --clustered --sparsity-ppm 11000 --depth 12 > $OUT/prof_det_lz4.log 2>&1
tools/prof_round.sh ${T}_l2_1pct --level 2 --sparsity-ppm 10000 > $OUT/prof_l2_1pct.log 2>&1
tools/prof_round.sh ${T}_l2_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 > $OUT/prof_l2_clustered.log 2>&1
else
tools/prof_round.sh ${T}_det_zstd --clustered --sparsity-ppm 11000 --depth 12 --scheme 1 > $OUT/prof_det_zstd.log 2>&1
tools/prof_round.sh ${T}_cfg5_b16 --config 5 --batch 16 --stack 32 > $OUT/prof_cfg5_b16.log 2>&1
tools/prof_round.sh ${T}_u32 --source-bytes 4 > $OUT/prof_u32.log 2>&1
tools/prof_round.sh ${T}_u8 --source-bytes 1 > $OUT/prof_u8.log 2>&1
tools/prof_round.sh ${T}_k2 --ny 3710 --nx 3838 --batch 64 --stack 128 > $OUT/prof_k2.log 2>&1
fi
echo done
