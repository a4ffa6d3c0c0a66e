#!/bin/bash
# round 5, experiment 39: by hand - the random-configuration tests with other seeds over the final tree (dense compaction, level 2 on two chains)
O=gpurun_out/r05_exp39.log
: > $O
for seed in 7001 7002; do
  RC_FUZZ_SEED=$seed RC_FUZZ_CASES=120 timeout -k 10 500 python3 -m pytest tests/test_gpu_api.py -x -q -m gpu -k "random_configurations" >> $O 2>&1; echo "small frames, seed $seed rc=$?" >> $O
  RC_FUZZ_SEED=$seed RC_FUZZ_BIG=1 RC_FUZZ_CASES=50 timeout -k 10 500 python3 -m pytest tests/test_gpu_api.py -x -q -m gpu -k "random_configurations" >> $O 2>&1; echo "frames of 30-260 tiles, seed $seed rc=$?" >> $O
  RC_FUZZ_SEED=$seed timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or l2" >> $O 2>&1; echo "parity fuzz + level 2, seed $seed rc=$?" >> $O
done
echo done >> $O
