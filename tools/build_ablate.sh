#!/bin/bash
# Profiling builds of the library with one kind of global store of the reduce kernel dropped (rc_reduce.hip, RC_ABLATE):
# ab_build/librecode_hip_abl<bits>.so for bits in "$@" (default 1 2 4 7).  Select with RC_AB_LIB in tools/quick_perf.py.
set -e
cd "$(dirname "$0")/../pyrecode_amd/csrc"
mkdir -p ../../ab_build
for b in ${@:-1 2 4 7}; do
  d=$(mktemp -d)
  for f in rc_api rc_reduce rc_lz4 rc_zstd rc_pix_huff rc_zstd_dec rc_blosc rc_l2 rc_expand; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -w -DRC_ABLATE=$b -c $f.hip -o $d/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_build/librecode_hip_abl$b.so $d/*.o
  rm -rf $d
done
ls -la ../../ab_build
