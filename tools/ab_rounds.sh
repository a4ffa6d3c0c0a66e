#!/bin/bash
# Same-box A/B of two TREES (e.g. last round's commit exported to ab_build/r03_tree with its own library built there, against this tree):
# tools/ab_rounds.sh <old tree>   - three interleaved rounds over the BASELINE configurations and the detector-like stacks
OLD=$1
run() { tree=$1; shift; (cd $tree && python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null) | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%9.0f fps  kernel %.4f  step %.4f  whole %.3f  rec %.0f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], j['config']['record_bytes_per_frame'], 'ok' if j['verified'] else 'NOT VERIFIED'), end='')"; }
while read -r cfg; do
  [ -z "$cfg" ] && continue
  for round in 1 2 3; do
    echo -n "$(printf '%-52s' "$cfg") | old: "; run $OLD $cfg; echo -n "  | new: "; run . $cfg; echo
  done
done <<CFGS
--config 2
--config 3
--config 4
--config 5
--config 5 --batch 16 --stack 32
--depth 12
--scheme 1 --depth 12
--clustered --sparsity-ppm 11000 --depth 12
--clustered --sparsity-ppm 11000 --depth 12 --scheme 1
--level 3
--scheme 0
--sparsity-ppm 100000 --stack 64
CFGS
