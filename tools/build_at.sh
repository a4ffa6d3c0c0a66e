#!/bin/bash
# Build librecode_hip from a git revision into ab_build/librecode_hip_<name>.so (same-box A/B runs: tools/ab.sh).
# usage: tools/build_at.sh <git-ref> <name>
set -e
REF=$1; NAME=$2
REPO=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d)
git -C $REPO archive $REF pyrecode_amd/csrc include | tar -x -C $d
cd $d/pyrecode_amd/csrc
SRCS=$(ls *.hip)
for f in $SRCS; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -w -c $f -o ${f%.hip}.o & done
wait
mkdir -p $REPO/ab_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/ab_build/librecode_hip_$NAME.so *.o
rm -rf $d
ls -la $REPO/ab_build/librecode_hip_$NAME.so
