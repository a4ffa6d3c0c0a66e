#!/bin/bash
# What each part of the reduce kernel costs at a given density: the same workload with the residual path dropped (--level 3), the
# codec dropped (--scheme 0: reduce-only records), both, and the kernel without the second stage next to it (--no-pipeline).
# usage: tools/decompose.sh <bench args of the workload...>
run() { python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'))"; }
for extra in "" "--no-pipeline" "--level 3" "--level 3 --no-pipeline" "--scheme 0" "--scheme 0 --no-pipeline" "--level 3 --scheme 0" "--level 3 --scheme 0 --no-pipeline"; do
  echo -n "$(printf '%-40s' "$extra") "; run "$@" $extra
done
