#!/usr/bin/env bash
# Build the REAL reference pieces that compile from their own sources, where they lie under
# /root/reference, into oracle/_ref/ (git-ignored; never copied into the repo as source).
#
#   c_recode*.so      <- /root/reference/pyrecode/pyrecode.cpp + c_extensions/reader.h
#                        (the reference's own CPython extension, setup.py:4-9 recipe: g++ -O3)
#   libreader_ref.so  <- oracle/ref_reader_shim.c, which only #includes the reference's
#                        c_extensions/reader.h in place and exports its three C functions
#                        unchanged (the CPython binding in pyrecode.cpp:121-141 mis-parses its
#                        arguments on LP64, SURVEY §0.4, so the C functions are also reached
#                        directly).
#   shim/numba        <- identity `jit` decorator, only so that `import pyrecode.recode_writer`
#                        succeeds in this container (numba is not installed); the reference's
#                        @jit kernels then run as the plain Python they are written in.
#                        Used ONLY by tests/golden/make_golden.py to capture fixtures.
#
# Test infrastructure only. Nothing in pyrecode_amd/ may load anything from here.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${RECODE_REFERENCE:-/root/reference}"
out="$here/_ref"
if [ ! -d "$ref/pyrecode" ]; then
  echo "build_ref: $ref not present (GPU box?) - keeping prebuilt files in $out" >&2
  exit 0
fi
mkdir -p "$out/shim/numba"
pyinc="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
ext="$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
g++ -shared -fPIC -O3 -I"$pyinc" -I"$ref/pyrecode/c_extensions" \
    "$ref/pyrecode/pyrecode.cpp" -o "$out/c_recode$ext"
gcc -shared -fPIC -O3 -DREF_READER_H="\"$ref/pyrecode/c_extensions/reader.h\"" \
    "$here/ref_reader_shim.c" -o "$out/libreader_ref.so"
cat > "$out/shim/numba/__init__.py" <<'EOF'
# identity-decorator stand-in (see oracle/build_ref.sh): the reference's @jit kernels run as plain Python
def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
njit = jit
prange = range
EOF
echo "build_ref: built $(ls "$out" | tr '\n' ' ')"
