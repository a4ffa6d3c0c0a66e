#!/bin/bash
O=gpurun_out/r05_exp16.log
: > $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp16_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp16_pytest.log >> $O
if grep -q "Aborted\|failed" gpurun_out/r05_exp16_pytest.log; then echo "stopping" >> $O; exit 1; fi
python3 bench.py --config 1 --no-cpu-baseline --no-ingest --min-seconds 1 > gpurun_out/final_r05a/cfg1.json 2>> gpurun_out/r05_exp16.err; tail -c 400 gpurun_out/final_r05a/cfg1.json >> $O
tools/final_profiles.sh r05a a >> $O 2>&1
echo done >> $O
