#!/bin/bash
# A sweep over the parameter space for slow paths (one short bench line each): tools/sweep_anomalies.sh
# (a line's stderr is kept - appended to gpurun_out/sweep_anomalies.err - so that an ERROR row says why)
run() { echo "== $*" >> gpurun_out/sweep_anomalies.err; timeout -k 10 120 python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.4 --no-cpu-baseline --no-ingest 2>> gpurun_out/sweep_anomalies.err | python3 -c "
import sys, json
try:
    j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-58s %9.0f fps  kernel %.4f  step %.4f  whole %.3f  rec %9.0f %s' % (sys.argv[1], j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], j['config']['record_bytes_per_frame'], 'ok' if j['verified'] else 'NOT VERIFIED'))
except Exception as e: print('%-58s ERROR %r' % (sys.argv[1], e))" "$*"; }
while read -r cfg; do [ -z "$cfg" ] && continue; run $cfg; done <<CFGS
--depth 9
--depth 10
--depth 14
--depth 15
--depth 10 --sparsity-ppm 50000
--depth 14 --scheme 1
--depth 8
--depth 1
--sparsity-ppm 0
--sparsity-ppm 100000 --stack 64 --batch 32
--sparsity-ppm 100
--sparsity-ppm 300000 --stack 64 --batch 32
--sparsity-ppm 600000 --stack 64 --batch 32
--sparsity-ppm 1000000 --stack 64 --batch 32 --scheme 0
--batch 63 --stack 252
--batch 65 --stack 260
--batch 1 --stack 64
--batch 7 --stack 63
--scheme 8
--scheme 8 --sparsity-ppm 50000 --depth 12
--level 2
--level 2 --scheme 1
--level 3 --scheme 1
--level 3 --scheme 8
--scheme 1 --clevel 3
--scheme 1 --clevel 2 --clustered --sparsity-ppm 11000 --depth 12
--scheme 0 --depth 12
--scheme 0 --level 3
CFGS
