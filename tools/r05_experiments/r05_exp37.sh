#!/bin/bash
# round 5, experiment 37: the final tree - smoke, the suite, the dense lines, the headline against round 4's tree once more, default bench, soak
O=gpurun_out/r05_exp37.log
: > $O
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids >> $O || exit 1
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_exp37_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp37_pytest.log >> $O
grep -q "rc=0" $O || exit 1
tools/ab_trees.sh ab_build/r04_tree 3 >> $O 2>&1 <<CFGS
--config 2
--sparsity-ppm 100000 --stack 64 --batch 32
--sparsity-ppm 300000 --stack 64 --batch 32
--config 5
CFGS
mkdir -p gpurun_out/final_r05c
Q="--no-cpu-baseline --no-ingest --min-seconds 1"
for l in "dense_10pct --sparsity-ppm 100000 --stack 64 --batch 32" "dense_30pct --sparsity-ppm 300000 --stack 64 --batch 32" "dense_60pct --sparsity-ppm 600000 --stack 32 --batch 16"; do
  set -- $l; n=$1; shift
  python3 bench.py "$@" $Q > gpurun_out/final_r05c/$n.json 2> gpurun_out/final_r05c/$n.err; tail -c 700 gpurun_out/final_r05c/$n.json | head -c 300 >> $O; echo >> $O
done
python3 bench.py > gpurun_out/final_r05c/default_bench.json 2> gpurun_out/final_r05c/default_bench.err; head -c 400 gpurun_out/final_r05c/default_bench.json >> $O; echo >> $O
timeout -k 10 400 python3 tools/soak_determinism.py 6 2>&1 | grep -v amdgpu.ids >> $O; echo "soak rc=$?" >> $O
echo done >> $O
