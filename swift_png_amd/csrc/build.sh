#!/bin/bash
# Builds libspng_mi355.so in-tree for gfx950 (cross-compiles without a GPU).  One object per source, compiled side by side
# (objects under build/, kept out of the history), linked into the shared library.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${SPNG_OUT:-../libspng_mi355.so}
OBJ=${SPNG_OBJ:-build}
SRCS="api.hip unfilter.hip inflate.hip pinflate2.hip encode.hip deflate.hip unpack.hip chunks.hip gzip.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${SPNG_EXTRA_FLAGS:-}"
HDRS="common.hpp huffman.hpp crc32.hpp ../../include/spng_mi355.h build.sh"
mkdir -p "$OBJ"
echo "$FLAGS" | cmp -s - "$OBJ/flags" || { rm -f "$OBJ"/*.o; echo "$FLAGS" > "$OBJ/flags"; }
newest_hdr=$(ls -t $HDRS | head -1)
pids=(); objs=()
for s in $SRCS; do
    o="$OBJ/${s%.hip}.o"; objs+=("$o")
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
        $HIPCC $FLAGS -c -o "$o" "$s" & pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -z "$p" ] || wait "$p"; done
newest_obj=$(ls -t "${objs[@]}" | head -1)
if [ ! -f "$OUT" ] || [ "$newest_obj" -nt "$OUT" ]; then
    $HIPCC --offload-arch=gfx950 -fPIC -shared -o "$OUT" "${objs[@]}"
fi
