#!/bin/bash
# round 5, experiment 26: k_zstd_fse with dynamic LDS (register claim 55 instead of 129), rows of 48 and 24 dwords, same box
O=gpurun_out/r05_exp26.log
: > $O
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "zstd" > gpurun_out/r05_exp26_pytest.log 2>&1; echo "pytest (zstd) rc=$?" >> $O; tail -n 2 gpurun_out/r05_exp26_pytest.log >> $O
grep -q "rc=0" $O || exit 1
for cfg in "--scheme 1" "--config 5" "--clustered --sparsity-ppm 11000 --depth 12 --scheme 1" "--config 3"; do
  python3 tools/ab_libs.py --rounds 3 --bench "$cfg" base=ab_build/librecode_hip_zbase.so dyn=ab_build/librecode_hip_zf.so dyn24=ab_build/librecode_hip_zf24.so >> $O 2>&1 || exit 1
done
echo done >> $O
