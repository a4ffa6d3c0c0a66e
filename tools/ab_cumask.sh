#!/bin/bash
# CU partition experiment: second stage on the first n CU-mask bits, reduce kernel on the others (RC_RSTREAM_EXCL) - tools/ab_cumask.sh
run() { python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'), end='')"; }
for cfg in "--config 2" "--config 3" "--clustered --sparsity-ppm 11000 --depth 12"; do
  for round in 1 2; do
    echo -n "$(printf '%-46s' "$cfg") | plain: "; run $cfg
    echo -n " | own stream: "; RC_BENCH_OWN_STREAM=1 run $cfg
    for n in 8 16 24 32 48; do
      echo -n " | excl $n: "; RC_BENCH_OWN_STREAM=1 RC_PSTREAM_CUS=$n RC_RSTREAM_EXCL=1 run $cfg
    done
    echo -n " | p-only 16: "; RC_PSTREAM_CUS=16 run $cfg
    echo
  done
done
