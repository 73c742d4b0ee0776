#!/bin/bash
# build_variant.sh NAME "EXTRA HIPCC FLAGS" [OBJECT ...] - a build variant of the library in stringzilla_amd/lib_variants/NAME/ (git-ignored,
# travels to the GPU box): the ordinary objects are copied, the named objects (default: all of them) are recompiled with the extra flags.
# Select it at run time with STRINGZILLAS_ROCM_LIBRARY=stringzilla_amd/lib_variants/NAME/libstringzillas_rocm_shared.so.
set -eu
NAME=$1; FLAGS=${2:-}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/stringzilla_amd/lib_variants/$NAME
make -s -C "$ROOT/stringzilla_amd/csrc" -j8
mkdir -p "$OUT"
rm -rf "$OUT/obj"; cp -r "$ROOT/stringzilla_amd/lib/obj" "$OUT/obj"
if [ $# -gt 0 ]; then for object in "$@"; do rm -f "$OUT/obj/$object.o"; done; else rm -f "$OUT"/obj/*.o; fi
find "$OUT/obj" -name "*.o" -exec touch {} +
make -s -C "$ROOT/stringzilla_amd/csrc" -j8 OUT="../lib_variants/$NAME" EXTRA="$FLAGS"
ls -la "$OUT/libstringzillas_rocm_shared.so"
