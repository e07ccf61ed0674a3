#!/bin/bash
# Builds libspng_mi355.so in-tree for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${SPNG_OUT:-../libspng_mi355.so}
SRCS="api.hip unfilter.hip inflate.hip pinflate2.hip encode.hip deflate.hip unpack.hip chunks.hip gzip.hip"
newest=$(ls -t $SRCS common.hpp huffman.hpp crc32.hpp ../../include/spng_mi355.h build.sh | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then exit 0; fi
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -o "$OUT" $SRCS ${SPNG_EXTRA_FLAGS:-}
