#!/bin/bash
# Same-box A/B of library builds over the DENSE configurations (BASELINE cfg 5, the detector-like stack) plus the headline as the
# "nothing loses" check: tools/ab_dense.sh libA.so libB.so ...   ("main" = the product build).  Three interleaved rounds.
run() { if [ $1 = main ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$(pwd)/$1; fi; shift
  python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'), end='')"; }
while read -r cfg; do
  [ -z "$cfg" ] && continue
  for round in 1 2 3; do
    echo -n "$(printf '%-52s' "$cfg")"
    for v in "$@"; do echo -n " | $(basename $v .so | sed s/librecode_hip_//): "; run $v $cfg; done
    echo
  done
done <<CFGS
--config 5
--clustered --sparsity-ppm 11000 --depth 12
--clustered --sparsity-ppm 11000 --depth 12 --scheme 1
--depth 12
--scheme 1 --depth 12
--config 2
CFGS
