#!/bin/bash
# What does each second-stage kernel cost the pipelined step?  tools/skip_bench.sh "<bench args>" <bits...>
# needs ab_build/librecode_hip_skip.so (tools/build_def.sh skip -DRC_DEV_SKIP); records are WRONG with any bit set ("verified": false)
ARGS=$1; shift
export RC_LIB_PATH=$(pwd)/ab_build/librecode_hip_skip.so
for round in 1 2; do
  for b in "$@"; do
    export RC_DEV_SKIP_BITS=$b
    echo -n "skip=$b: "
    python3 bench.py $ARGS --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
r = j['roofline']
print('%.4f ms/step  reduce kernel %.4f ms' % (j['ms_per_step'], r.get('kernel_ms', 0)))"
  done
done
