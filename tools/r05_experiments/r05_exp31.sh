#!/bin/bash
# round 5, experiment 31: the tree with level 2 in its final form - smoke, the suite, level 2 against round 4's tree on the same box, the
# level-2 chain kernel by kernel, bench lines, determinism soak
O=gpurun_out/r05_exp31.log
: > $O
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O 2>&1 || exit 1
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_exp31_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp31_pytest.log >> $O
grep -q "rc=0" $O || exit 1
tools/ab_trees.sh ab_build/r04_tree 3 >> $O 2>&1 <<CFGS
--level 2 --sparsity-ppm 10000
--level 2 --clustered --sparsity-ppm 2000 --depth 12
--config 4
--config 2
CFGS
tools/prof_bench.sh r05_l2f_1pct --level 2 --sparsity-ppm 10000 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2f_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2f_cfg4 --config 4 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
mkdir -p gpurun_out/final_r05b
Q="--no-cpu-baseline --no-ingest --min-seconds 1"
for l in "l2_1pct --level 2 --sparsity-ppm 10000" "l2_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12" "cfg4 --config 4" "l2_10pct --level 2 --sparsity-ppm 100000 --stack 64 --batch 32"; do
  set -- $l; n=$1; shift
  python3 bench.py "$@" $Q > gpurun_out/final_r05b/$n.json 2> gpurun_out/final_r05b/$n.err; tail -c 700 gpurun_out/final_r05b/$n.json | head -c 300 >> $O; echo >> $O
done
python3 bench.py > gpurun_out/final_r05b/default_bench.json 2> gpurun_out/final_r05b/default_bench.err; cat gpurun_out/final_r05b/default_bench.json >> $O
timeout -k 10 400 python3 tools/soak_determinism.py 8 >> $O 2>&1; echo "soak rc=$?" >> $O
echo done >> $O
