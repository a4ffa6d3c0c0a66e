#!/bin/bash
# 15 against 16 waves per CU: the product build with three-wave workgroups against a build with a 32-value compaction stage (10.2 KB of LDS per
# wave) and four-wave workgroups, on data sparse enough for that stage (0.2 %)
run() { python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], 'ok' if j['verified'] else 'NOT VERIFIED'), end='')"; }
for cfg in "--sparsity-ppm 2000" "--sparsity-ppm 2000 --depth 12"; do
  for round in 1 2 3; do
    echo -n "$(printf '%-36s' "$cfg") | product (3 waves x 5): "; run $cfg
    echo -n " | product, 4 waves x 3: "; RC_REDUCE_WG_WAVES=4 run $cfg
    echo -n " | cap32, 4 waves x 4: "; RC_LIB_PATH=$(pwd)/ab_build/librecode_hip_cap32.so RC_REDUCE_WG_WAVES=4 run $cfg
    echo -n " | cap32, 3 waves x 5: "; RC_LIB_PATH=$(pwd)/ab_build/librecode_hip_cap32.so RC_REDUCE_WG_WAVES=3 run $cfg
    echo
  done
done
