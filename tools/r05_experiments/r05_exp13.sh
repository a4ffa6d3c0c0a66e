#!/bin/bash
# round 5, experiment 13: round 4's tree against this one, same box
O=gpurun_out/r05_exp13.log
: > $O
tools/ab_trees.sh ab_build/r04_tree 3 >> $O 2>&1 <<CFGS
--config 2
--config 3
--config 4
--config 5
--depth 12
--clustered --sparsity-ppm 11000 --depth 12
--clustered --sparsity-ppm 11000 --depth 12 --scheme 1
--level 2 --sparsity-ppm 10000
--level 2 --clustered --sparsity-ppm 2000 --depth 12
--level 3
--scheme 0
--sparsity-ppm 100000 --stack 64 --batch 32
CFGS
echo done >> $O
