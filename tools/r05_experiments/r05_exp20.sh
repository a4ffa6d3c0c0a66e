#!/bin/bash
# round 5, experiment 20: does the level-2 stage cost the reduce kernel LDS room (its one-wave workgroups hold 12 - 17 KB each)?
# persistent grids (fewer resident workgroups) and extra dynamic LDS per workgroup, same box
O=gpurun_out/r05_exp20.log
D=ab_build/librecode_hip_dev.so
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --sparsity-ppm 10000" base=$D w256=$D,RC_L2_WGS=256 w512=$D,RC_L2_WGS=512 w1024=$D,RC_L2_WGS=1024 w2048=$D,RC_L2_WGS=2048 pad16k=$D,RC_L2_DYNLDS=16384 pad40k=$D,RC_L2_DYNLDS=40000 > $O 2>&1 || exit 1
python3 tools/ab_libs.py --rounds 2 --bench "--level 2 --clustered --sparsity-ppm 2000 --depth 12" base=$D w512=$D,RC_L2_WGS=512 w1024=$D,RC_L2_WGS=1024 w2048=$D,RC_L2_WGS=2048 pad40k=$D,RC_L2_DYNLDS=40000 >> $O 2>&1 || exit 1
python3 tools/ab_libs.py --rounds 2 --bench "--config 4" base=$D w512=$D,RC_L2_WGS=512 w1024=$D,RC_L2_WGS=1024 w2048=$D,RC_L2_WGS=2048 >> $O 2>&1
echo done >> $O
