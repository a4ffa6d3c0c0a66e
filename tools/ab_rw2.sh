#!/bin/bash
# workgroups of two waves (x 7 per CU) against three (x 5): tools/ab_rw2.sh   (ab_build/librecode_hip_rw2.so: -DRC_RW_ALT=2)
run() { python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'), end='')"; }
for cfg in "--config 2" "--config 3" "--config 5" "--clustered --sparsity-ppm 11000 --depth 12" "--config 4"; do
  for round in 1 2 3; do
    echo -n "$(printf '%-46s' "$cfg") | product: "; run $cfg
    echo -n " | forced 3: "; RC_REDUCE_WG_WAVES=3 run $cfg
    echo -n " | forced 2: "; RC_LIB_PATH=$(pwd)/ab_build/librecode_hip_rw2.so RC_REDUCE_WG_WAVES=3 run $cfg
    echo
  done
done
