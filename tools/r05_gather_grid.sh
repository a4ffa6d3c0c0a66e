#!/bin/bash
# round 5: k_gather against k_assemble (RC_OLD_ASSEMBLE=1) and over its grid size, same box, interleaved (a -DRC_DEV_KNOBS build)
# usage: tools/r05_gather_grid.sh <lib> <out log> <bench args...>
LIB=$1; O=$2; shift; shift
export RC_LIB_PATH=$(pwd)/$LIB
run() { python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f  rec %.0f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], j['config']['record_bytes_per_frame'], 'ok' if j['verified'] else 'NOT VERIFIED'))"; }
echo "== $LIB | $*" >> $O
for round in 1 2; do
  for v in "RC_OLD_ASSEMBLE=1" "RC_GATHER_WGS=128" "RC_GATHER_WGS=256" "RC_GATHER_WGS=512" "RC_GATHER_WGS=1024" "RC_GATHER_WGS=0"; do
    echo -n "$(printf '%-22s' $v) " >> $O
    env $v bash -c "$(declare -f run); run $*" >> $O 2>&1
  done
done
