#!/bin/bash
# round 5, experiment 3: k_gather variants (rounds in flight, registers), reduce-kernel priority, stand-alone stage times, kernel trace
O=gpurun_out/r05_exp3.log
: > $O
A=ab_build/librecode_hip
python3 tools/ab_libs.py --rounds 4 old=${A}_g4.so,RC_OLD_ASSEMBLE=1 g4=${A}_g4.so g8=${A}_g8.so g12=${A}_g12.so g2w8=${A}_g2w8.so g4w8=${A}_g4w8.so prio2=${A}_g4prio2.so >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--clustered --sparsity-ppm 11000 --depth 12" old=${A}_g4.so,RC_OLD_ASSEMBLE=1 g4=${A}_g4.so g8=${A}_g8.so g12=${A}_g12.so g4w8=${A}_g4w8.so prio2=${A}_g4prio2.so >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--config 5" old=${A}_g4.so,RC_OLD_ASSEMBLE=1 g4=${A}_g4.so g8=${A}_g8.so g12=${A}_g12.so g4w8=${A}_g4w8.so prio2=${A}_g4prio2.so >> $O 2>&1
python3 tools/ab_libs.py --rounds 3 --bench "--config 3" old=${A}_g4.so,RC_OLD_ASSEMBLE=1 g4=${A}_g4.so g8=${A}_g8.so g12=${A}_g12.so >> $O 2>&1
echo "== stand-alone stage times (quick_perf: reduce, codec, scan, layout+assemble, total)" >> $O
for v in "RC_OLD_ASSEMBLE=1 RC_AB_LIB=${A}_g4.so" "RC_AB_LIB=${A}_g4.so" "RC_AB_LIB=${A}_g8.so" "RC_AB_LIB=${A}_g12.so"; do
  echo "-- $v" >> $O
  env $v python3 tools/quick_perf.py 4096 4096 64 10000 16 2 2>&1 | grep shape >> $O
  env $v python3 tools/quick_perf.py 4096 4096 64 10000 12 2 2>&1 | grep shape >> $O
  env $v python3 tools/quick_perf.py 8184 11520 32 50000 12 1 2>&1 | grep shape >> $O
done
echo "== kernel trace of the pipelined bench (product build)" >> $O
tools/prof_bench.sh r05_exp3_lz4 --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline --no-ingest >> $O 2>&1
echo done >> $O
