#!/bin/bash
# round 5, experiment 4: the new tests, level-2 kernel traces, the dense regime (sweep rows that used to read ERROR, decompose at 10 % / 30 %)
O=gpurun_out/r05_exp4.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench.py -m gpu -x -q -k "dense or all_set or rccl or torchrun" > gpurun_out/r05_exp4_pytest1.log 2>&1; echo "pytest1 rc=$?" >> $O; tail -n 3 gpurun_out/r05_exp4_pytest1.log >> $O
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "failed_run or random_config or coo" > gpurun_out/r05_exp4_pytest2.log 2>&1; echo "pytest2 rc=$?" >> $O; tail -n 3 gpurun_out/r05_exp4_pytest2.log >> $O
echo "== level 2 kernel traces" >> $O
tools/prof_bench.sh r05_l2_1pct --level 2 --sparsity-ppm 10000 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_cfg4 --config 4 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo "== level 2 lines" >> $O
for a in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--config 4"; do
  python3 bench.py $a --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-60s %9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (sys.argv[1], j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'))" "$a" >> $O
done
echo "== dense: decompose 10 %" >> $O
tools/decompose.sh --sparsity-ppm 100000 --stack 64 --batch 32 >> $O 2>&1
echo "== dense: decompose 30 %" >> $O
tools/decompose.sh --sparsity-ppm 300000 --stack 64 --batch 32 >> $O 2>&1
echo "== sweep" >> $O
rm -f gpurun_out/sweep_anomalies.err
tools/sweep_anomalies.sh >> $O 2>&1
echo done >> $O
