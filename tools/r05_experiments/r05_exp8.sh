#!/bin/bash
# round 5, experiment 6: the new level-2 stage - its tests, then the whole suite, bench lines and kernel traces
O=gpurun_out/r05_exp8.log
: > $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "l2 or level_2 or config4 or random_config or uint8" > gpurun_out/r05_exp8_pytest1.log 2>&1; echo "pytest (level 2) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp8_pytest1.log >> $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp8_pytest2.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp8_pytest2.log >> $O
echo "== level 2 lines" >> $O
for a in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--config 4" "--level 2 --scheme 1" "--level 2 --source-bytes 1" "--level 2 --sparsity-ppm 50000 --depth 12"; do
  python3 bench.py $a --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>>gpurun_out/r05_exp8.err | python3 -c "
import sys, json
try:
    j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-60s %9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (sys.argv[1], j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'))
except Exception as e: print(sys.argv[1], 'ERROR', repr(e))" "$a" >> $O
done
echo "== level 2 kernel traces" >> $O
tools/prof_bench.sh r05_l2v2_1pct --level 2 --sparsity-ppm 10000 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2v2_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2v2_cfg4 --config 4 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo "== dense after the LZ4 store-without-parse rule" >> $O
tools/decompose.sh --sparsity-ppm 100000 --stack 64 --batch 32 >> $O 2>&1
echo done >> $O
