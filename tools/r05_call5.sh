#!/bin/bash
# round 5, fifth GPU call: where the search kernel's cycles go on compressible input (SPNG_D3_PROF build: variants/libspng_d3prof.so),
# then the product build's probes (nested loop again, the heap replay two levels per round trip)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
SPNG_LIB=/root/repo/variants/libspng_d3prof.so PROBE_LEVEL=6 PROBE_WHICH=synth4k PROBE_N=4 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05e_d3prof_l6.log 2>&1; grep -E "d3 prof" gpurun_out/r05e_d3prof_l6.log | sort -t'(' -k2 | awk 'NR%12==1' | head -40; grep "streams," gpurun_out/r05e_d3prof_l6.log
SPNG_LIB=/root/repo/variants/libspng_d3prof.so PROBE_LEVEL=9 PROBE_WHICH=photo PROBE_N=8 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05e_d3prof_l9.log 2>&1; grep -E "d3 prof" gpurun_out/r05e_d3prof_l9.log | awk 'NR%6==1' | head -16; grep "streams," gpurun_out/r05e_d3prof_l9.log
PROBE_WHICH=photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/r05e_probe_l9_256.log 2>&1; tail -1 gpurun_out/r05e_probe_l9_256.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,random PROBE_N=256 timeout 400 python tools/probe_deflate2.py > gpurun_out/r05e_probe_l6_256.log 2>&1; tail -2 gpurun_out/r05e_probe_l6_256.log
