#!/bin/bash
# Builds the library of a git revision (or of the working tree: rev = WORK) into variants/libspng_<name>.so -- the A/B legs of a
# tuning call (SPNG_LIB=variants/libspng_<name>.so python tools/probe_v2.py ...).  variants/ is git-ignored and travels with gpurun.
#   tools/build_variant.sh <rev|WORK> <name> [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")/.."
rev=$1; name=$2; shift 2
mkdir -p variants
out="$PWD/variants/libspng_${name}.so"
if [ "$rev" = WORK ]; then
    SPNG_OUT="$out" SPNG_OBJ="$PWD/variants/obj_${name}" SPNG_EXTRA_FLAGS="$*" swift_png_amd/csrc/build.sh
else
    tmp=$(mktemp -d); trap 'rm -rf "$tmp"' EXIT
    git archive "$rev" swift_png_amd/csrc include | tar -x -C "$tmp"
    SPNG_OUT="$out" SPNG_OBJ="$PWD/variants/obj_${name}" SPNG_EXTRA_FLAGS="$*" "$tmp/swift_png_amd/csrc/build.sh"
fi
ls -la "$out"
