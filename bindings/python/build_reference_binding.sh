#!/bin/bash
# build_reference_binding.sh - compiles the REFERENCE's own CPython modules from the sources where they lie under
# $REFERENCE (nothing is copied into this repository), following the reference's setup.py source lists:
#   stringzilla   : python/stringzilla/*.c + c/stringzilla/*.c                      (the base module: Str, Strs, ...)
#   stringzillas  : python/stringzillas/*.c, linked against OUR libstringzillas_rocm_shared.so instead of the
#                   reference's c/stringzillas/*.cpp|.cu shim units - i.e. the `stringzillas-rocm` target the reference
#                   declares in setup.py:863-865 but never defines.
# This IS the drop-in at the Python level: the reference's binding sources, unmodified, over this library's C-ABI.
#   usage: build_reference_binding.sh [OUT_DIR]      (REFERENCE=/path/to/StringZilla, default /root/reference)
# CMake drives it as the optional target `stringzillas_rocm_python` (-DSTRINGZILLAS_ROCM_REFERENCE_ROOT=...); the test
# suite calls it with OUT_DIR = oracle/_ref/pybinding (git-ignored, travels to the GPU box with the snapshot), where
# tests/test_reference_binding.py and tests/test_reference_suite.py drive the product through the reference's own Python API.
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
REFERENCE=${REFERENCE:-/root/reference}
OUT=${1:-$ROOT/oracle/_ref/pybinding}
LIBDIR=${STRINGZILLAS_ROCM_LIBDIR:-$ROOT/stringzilla_amd/lib}
if [ ! -d "$REFERENCE/python/stringzillas" ]; then echo "reference tree $REFERENCE absent: keeping prebuilt $OUT (if any)"; exit 0; fi
mkdir -p "$OUT/obj"
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
SUFFIX=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
MACROS="-DSZ_DYNAMIC_DISPATCH=1 -DSZ_IS_BIG_ENDIAN_=0 -DSZ_IS_64BIT_X86_=1 -DSZ_IS_64BIT_ARM_=0 -DSZ_USE_WESTMERE=1 -DSZ_USE_GOLDMONT=1 -DSZ_USE_HASWELL=1 -DSZ_USE_SKYLAKE=1 -DSZ_USE_ICELAKE=1 -DSZ_USE_NEON=0 -DSZ_USE_NEONAES=0 -DSZ_USE_NEONSHA=0 -DSZ_USE_SVE=0 -DSZ_USE_SVE2=0 -DSZ_USE_SVE2AES=0"
CFLAGS="-std=c99 -D_GNU_SOURCE -O2 -fPIC -w"

compile() { # compile NAME SOURCE... -> objects in $OUT/obj/NAME_*.o, in parallel
    local name=$1; shift
    local pids=()
    for source in "$@"; do
        local object="$OUT/obj/${name}_$(echo "$source" | tr '/' '_' | sed 's/\.c$/.o/')"
        gcc $CFLAGS $MACROS $EXTRA -I"$REFERENCE/include" -I"$PYINC" -I"$NPINC" -c "$REFERENCE/$source" -o "$object" &
        pids+=($!)
        if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
    done
    wait
}

if [ ! -f "$OUT/stringzilla$SUFFIX" ]; then
    EXTRA="-I$REFERENCE/c/stringzilla"
    compile base $(cd "$REFERENCE" && ls python/stringzilla/*.c c/stringzilla/*.c)
    gcc -shared -fPIC "$OUT"/obj/base_*.o -o "$OUT/stringzilla$SUFFIX"
fi
EXTRA="-I$REFERENCE/c/stringzillas -DSZ_USE_CUDA=1 -DFU_WITH_TOPOLOGY=0"
compile szs python/stringzillas/stringzillas.c python/stringzillas/device_scope.c python/stringzillas/similarities.c python/stringzillas/fingerprints.c
gcc -shared -fPIC "$OUT"/obj/szs_*.o -o "$OUT/stringzillas$SUFFIX" -L"$LIBDIR" -lstringzillas_rocm_shared \
    -Wl,-rpath,"$LIBDIR" -Wl,-rpath,'$ORIGIN/../../../stringzilla_amd/lib'
rm -rf "$OUT/obj"
# The reference's OWN Python test suite for this path (test/similarities.py and the two helper modules it imports) is
# carried along the same way - into oracle/_ref/, which is git-ignored: reference-derived, never in this repository's
# history, but it travels to the GPU box with the snapshot, where /root/reference does not exist.
# tests/test_reference_suite.py runs it, unmodified, against the binding built above.
if [ "$OUT" = "$ROOT/oracle/_ref/pybinding" ]; then
    SUITE=$ROOT/oracle/_ref/reference_tests/test
    mkdir -p "$SUITE"
    for file in __init__.py similarities.py sz_helpers.py szs_helpers.py; do cp "$REFERENCE/test/$file" "$SUITE/$file"; done
fi
echo "built $OUT/stringzilla$SUFFIX and $OUT/stringzillas$SUFFIX against libstringzillas_rocm_shared.so"
