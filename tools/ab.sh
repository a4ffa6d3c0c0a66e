#!/bin/bash
# Same-box A/B of library builds with tools/quick_perf.py: tools/ab.sh "<quick_perf args>" libA.so libB.so ...  ("main" = the product build)
# three interleaved rounds; prints one line per (round, build).
ARGS=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    if [ $v = main ]; then unset RC_AB_LIB; else export RC_AB_LIB=$(pwd)/$v; fi
    echo -n "$(basename $v): "
    python3 tools/quick_perf.py $ARGS 2>&1 | grep shape | sed 's/.*median ms/median ms/'
  done
done
