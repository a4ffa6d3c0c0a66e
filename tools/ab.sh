#!/bin/bash
# A/B on one box: ab_build/librecode_hip_A.so (a build of an earlier commit) against the in-tree library, interleaved.
# usage: tools/ab.sh <quick_perf args...>
for i in $(seq 1 ${AB_N:-3}); do
  echo "A:"; RC_AB_LIB=ab_build/librecode_hip_A.so timeout -k 10 120 python tools/quick_perf.py "$@" || exit 1
  echo "B:"; timeout -k 10 120 python tools/quick_perf.py "$@" || exit 1
done
