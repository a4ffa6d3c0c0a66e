#!/bin/bash
# Same-box A/B of two library builds over several bench.py configurations: tools/ab_configs.sh libA.so libB.so   ("main" = the product build)
# Two interleaved rounds per configuration; prints frames/s and the reduce kernel's ms per build.
A=$1; B=$2
run() { if [ $1 = main ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$(pwd)/$1; fi; shift
  python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f ms  step %.4f ms' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step']), end='')"; }
while read -r cfg; do
  [ -z "$cfg" ] && continue
  for round in 1 2; do
    echo -n "$(printf '%-46s' "$cfg") | $A: "; run $A $cfg; echo -n "  | $B: "; run $B $cfg; echo
  done
done <<CFGS
--config 2
--config 3
--config 5
--config 5 --batch 16 --stack 32
--depth 12
--config 4
--scheme 0
--level 3
--sparsity-ppm 100000 --stack 64
--clustered --sparsity-ppm 11000 --depth 12
CFGS
