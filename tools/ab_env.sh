#!/bin/bash
# Same-box A/B of one build under two environments: tools/ab_env.sh "<VAR=value>" <bench args...>   (three interleaved rounds; A = with the variable)
V=$1; shift
run() { python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f  rec %.0f B %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], j['config']['record_bytes_per_frame'], 'ok' if j['verified'] else 'NOT VERIFIED'), end='')"; }
for round in 1 2 3; do
  echo -n "$(printf '%-44s' "$*") | $V: "; env $V bash -c "$(declare -f run); run $*"; echo -n "  | default: "; run "$@"; echo
done
