#!/bin/bash
# Same-box A/B of library builds under bench.py (pipelined mode, the driver's view): tools/ab_bench.sh "<bench args>" libA.so libB.so ...
# ("main" = the product build).  Three interleaved rounds; prints value, ms/step, reduce-kernel ms and whole_path_frac per (round, build).
ARGS=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    if [ $v = main ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$(pwd)/$v; fi
    echo -n "$(basename $v): "
    python3 bench.py $ARGS --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
r = j['roofline']
print('%.0f %s  %.4f ms/step  kernel %.4f ms  whole_path_frac %.3f  frac %.3f' % (j['value'], j['unit'], j['ms_per_step'], r.get('kernel_ms', 0), r.get('whole_path_frac', 0), r['frac']))"
  done
done
