#!/bin/bash
# round 5, experiment 38: PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the level-2 configurations in their final form
O=gpurun_out/r05_exp38.log
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "l2 or level2 or level_2 or bench" > $O 2>&1; echo "pytest (l2, bench) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
T=r05d
tools/prof_round.sh ${T}_cfg4 --config 4 >> $O 2>&1
tools/prof_round.sh ${T}_l2_1pct --level 2 --sparsity-ppm 10000 >> $O 2>&1
tools/prof_round.sh ${T}_l2_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 >> $O 2>&1
echo done >> $O
