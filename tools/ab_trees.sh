#!/bin/bash
# Same-box A/B of two TREES: an older commit exported to <old tree> with its own library built there (git archive <commit> | tar -x -C <old tree>;
# make -C <old tree>/pyrecode_amd/csrc; make -C <old tree>/oracle) against this tree - interleaved rounds per configuration, medians at the end.
# usage: tools/ab_trees.sh <old tree> <rounds> <<< "one bench.py argument line per configuration"
OLD=$1; R=${2:-3}
run() { tree=$1; shift; (cd $tree && python3 bench.py "$@" --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest 2>/dev/null) | python3 -c "
import sys, json
try:
    j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.0f %.4f %.4f %.3f %.0f %s' % (j['value'], j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['whole_path_frac'], j['config']['record_bytes_per_frame'], 'ok' if j['verified'] else 'NOT-VERIFIED'))
except Exception as e: print('0 0 0 0 0 ERROR')"; }
while read -r cfg; do
  [ -z "$cfg" ] && continue
  run . $cfg > /dev/null     # warm-up
  o=""; n=""
  for round in $(seq 1 $R); do
    if [ $((round % 2)) = 1 ]; then o="$o$(run $OLD $cfg)\n"; n="$n$(run . $cfg)\n"; else n="$n$(run . $cfg)\n"; o="$o$(run $OLD $cfg)\n"; fi
  done
  python3 - "$cfg" "$o" "$n" <<'PY'
import sys, statistics as st
cfg, o, n = sys.argv[1], sys.argv[2], sys.argv[3]
def parse(t):
    rows = [l.split() for l in t.replace("\\n", "\n").splitlines() if l.strip()]
    return rows
ro, rn = parse(o), parse(n)
mo, mn = st.median(float(r[0]) for r in ro), st.median(float(r[0]) for r in rn)
print("%-58s | old %8.0f fps (whole %.3f, rec %s)  new %8.0f fps (whole %.3f, kernel %.4f, step %.4f, rec %s)  %+5.1f %%  %s | old: %s | new: %s" % (
    cfg, mo, st.median(float(r[3]) for r in ro), ro[0][4], mn, st.median(float(r[3]) for r in rn), st.median(float(r[1]) for r in rn), st.median(float(r[2]) for r in rn), rn[0][4],
    100 * (mn / mo - 1) if mo else 0, "ok" if all(r[5] == "ok" for r in ro + rn) else "CHECK", " ".join(r[0] for r in ro), " ".join(r[0] for r in rn)), flush=True)
PY
done
