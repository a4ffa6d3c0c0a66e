#!/bin/bash
# round 5, experiment 9: level 2 with resting nodes; LZ4 emit / dense compaction / event-wait changes: whole suite, A/B against the g4 build, phase shares
O=gpurun_out/r05_exp9.log
: > $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp9_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp9_pytest.log >> $O
echo "== level 2 lines" >> $O
for a in "--level 2 --sparsity-ppm 10000" "--level 2 --clustered --sparsity-ppm 2000 --depth 12" "--config 4"; do
  python3 bench.py $a --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>>gpurun_out/r05_exp9.err | python3 -c "
import sys, json
try:
    j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-60s %9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (sys.argv[1], j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'))
except Exception as e: print(sys.argv[1], 'ERROR', repr(e))" "$a" >> $O
done
A=ab_build/librecode_hip
python3 tools/ab_libs.py --rounds 4 g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--clustered --sparsity-ppm 11000 --depth 12" g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--config 3" g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--config 5" g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--sparsity-ppm 100000 --stack 64 --batch 32" g4=${A}_g4.so main >> $O 2>&1
python3 tools/ab_libs.py --rounds 2 --bench "--sparsity-ppm 300000 --stack 64 --batch 32" g4=${A}_g4.so main >> $O 2>&1
echo "== phase shares" >> $O
for a in "4096 4096 32 10000 16 2" "4096 4096 32 100000 16 2" "4096 4096 32 300000 16 2"; do
  RC_AB_LIB=ab_build/librecode_hip_phase.so python3 tools/phase_timing.py $a 2>&1 | grep -v amdgpu.ids >> $O
done
tools/prof_bench.sh r05_l2v3_cfg4 --config 4 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo done >> $O
