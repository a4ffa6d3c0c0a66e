#!/bin/bash
# round 5, experiment 32: the second-stage stream at low / high priority (does the dispatcher then give freed CU room back to the reduce
# kernel first?), same box; and the new two-chain failure test
O=gpurun_out/r05_exp32.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "async" > $O 2>&1; echo "pytest (async) rc=$?" >> $O
grep -q "rc=0" $O || exit 1
D=ab_build/librecode_hip_prio.so
for cfg in "" "--scheme 1" "--config 5" "--clustered --sparsity-ppm 11000 --depth 12"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" base=$D low=$D,RC_PSTREAM_PRIO=2 high=$D,RC_PSTREAM_PRIO=1 >> $O 2>&1 || exit 1
done
echo done >> $O
