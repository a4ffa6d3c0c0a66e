#!/bin/bash
# round 5, experiment 18: the tree as committed - the driver's own sequence (build record, smoke, default bench), the suite, determinism soak
O=gpurun_out/r05_exp18.log
: > $O
python3 -c "import __graft_entry__ as g; g.smoke()" >> $O 2>&1; echo "smoke rc=$?" >> $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_exp18_pytest.log 2>&1; echo "pytest (all) rc=$?" >> $O; tail -n 4 gpurun_out/r05_exp18_pytest.log >> $O
python3 bench.py > gpurun_out/r05_exp18_bench_default.json 2> gpurun_out/r05_exp18_bench_default.err; echo "bench rc=$?" >> $O; python3 -c "
import json
j=json.loads([l for l in open('gpurun_out/r05_exp18_bench_default.json').read().splitlines() if l.startswith('{')][-1])
print('default bench: %.0f frames/s  %.4f ms/step  frac %.4f whole %.4f verified %s records_checked %d cpu_baseline %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['whole_path_frac'], j['verified'], len(j['records_checked']), j['cpu_baseline']['value']))" >> $O 2>&1
timeout -k 10 400 python3 tools/soak_determinism.py 8 >> $O 2>&1; echo "soak rc=$?" >> $O
echo done >> $O
