#!/bin/bash
# round 5, fourteenth GPU call: A/B only -- decode's token loops with one way out (V1: a lane that stops parks its position behind
# the loop's bound instead of `break`) and with the distance lookup outside a branch in rounds 0 and 1 (V2); the shipped build twice
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in i_ship i_v1 i_v2 i_v12 i_ship i_v12; do
  SPNG_LIB=/root/repo/variants/libspng_$v.so timeout 200 python tools/probe_v2.py --kinds swiftpng --steps 4 > gpurun_out/r05q_ab_$v.log 2>&1
  echo "== $v $(grep -E '^swiftpng auto' gpurun_out/r05q_ab_$v.log | cut -c1-200)"
done
