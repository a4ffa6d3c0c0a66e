#!/bin/bash
# Same-box A/B of library builds over the LZ4 configurations (headline, dense maps): tools/ab_lz4.sh libA.so libB.so ("main" = the product build)
run() { if [ $1 = main ]; then unset RC_LIB_PATH; else export RC_LIB_PATH=$(pwd)/$1; fi; shift
  python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f  rec %.0f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], j['config']['record_bytes_per_frame'], 'ok' if j['verified'] else 'NOT VERIFIED'), end='')"; }
while read -r cfg; do
  [ -z "$cfg" ] && continue
  for round in 1 2 3; do
    echo -n "$(printf '%-46s' "$cfg")"
    for v in "$@"; do echo -n " | $(basename $v .so | sed s/librecode_hip_//): "; run $v $cfg; done
    echo
  done
done <<CFGS
--clustered --sparsity-ppm 11000 --depth 12
--sparsity-ppm 20000 --depth 12
--sparsity-ppm 15000
--config 2
--depth 12
--level 3
CFGS
