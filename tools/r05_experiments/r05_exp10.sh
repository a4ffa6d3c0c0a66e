#!/bin/bash
# round 5, experiment 10: level 2 - directory through the results store, neighbours' words through DPP
O=gpurun_out/r05_exp10.log
: > $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp10_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp10_pytest.log >> $O
echo "== level 2 lines" >> $O
for a in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--config 4" "--config 4" "--level 2 --scheme 1"; do
  python3 bench.py $a --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>>gpurun_out/r05_exp10.err | python3 -c "
import sys, json
try:
    j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-60s %9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (sys.argv[1], j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'))
except Exception as e: print(sys.argv[1], 'ERROR', repr(e))" "$a" >> $O
done
tools/prof_bench.sh r05_l2v4_cfg4 --config 4 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
tools/prof_bench.sh r05_l2v4_1pct --level 2 --sparsity-ppm 10000 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
A=ab_build/librecode_hip
python3 tools/ab_libs.py --rounds 3 g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--clustered --sparsity-ppm 11000 --depth 12" g4=${A}_g4.so main >> $O 2>&1
echo done >> $O
