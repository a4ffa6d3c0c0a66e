#!/bin/bash
# A build of the library with extra -D flags: tools/build_def.sh <name> <flags...>  ->  ab_build/librecode_hip_<name>.so
# (select with RC_LIB_PATH, see pyrecode_amd/_lib.py; tools/ab_bench.sh runs bench.py over several builds on one box)
set -e
name=$1; shift
cd "$(dirname "$0")/../pyrecode_amd/csrc"
mkdir -p ../../ab_build
d=$(mktemp -d)
# the file list is the Makefile's (a build that lacks a translation unit lacks its symbols, and _lib.lib() binds them all)
for f in $(sed -n 's/^SRCS := //p' Makefile | sed 's/\.hip//g'); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -w "$@" -c $f.hip -o $d/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_build/librecode_hip_$name.so $d/*.o -ldl
rm -rf $d
ls -la ../../ab_build/librecode_hip_$name.so
