#!/bin/bash
# Round-end measurement on the GPU box: parity suite, smoke, kernel-trace stats of the decode and of the encode step, the PMC passes
# (decode: FETCH_SIZE / WRITE_SIZE of the headline workload; encode: the same two counters + instruction counters of the level-9
# deflate of 1024 random 64 MiB streams), the N = 8 shard shape, then the headline bench line (it reads the traffic measured here).
# Everything lands in gpurun_out/ (copied to profiles/ afterwards).
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
export R=${R:-r05}
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $1"; }
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
# A/B in front of everything (AB="name name": builds under variants/, tools/build_variant.sh): the shipped library is the LAST name's
# build; an earlier one that decodes the headline workload more than 0.5 % faster takes its place for the rest of this run
# (gpurun_out/${R}_ab.txt says which -- the source is then set to match before the round ends)
if [ -n "$AB" ]; then
  for v in $AB; do
    SPNG_LIB=/root/repo/variants/libspng_$v.so timeout 300 python tools/probe_v2.py --kinds swiftpng --steps 3 > gpurun_out/${R}_ab_$v.log 2>&1
    echo "== $v $(grep -E '^swiftpng auto' gpurun_out/${R}_ab_$v.log | cut -c1-230)"
  done
  python tools/ab_pick.py $R $AB > gpurun_out/${R}_ab.txt
  cat gpurun_out/${R}_ab.txt; lap "A/B"
fi
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${R}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${R}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; tail -1 gpurun_out/${R}_smoke.log; lap "tests + smoke"
# decode: kernel-trace stats of the bench command
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 3 --warmup 1 --no-swiftpng --no-cpu-baseline --no-extras > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/prof_stats.err
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_rocprof_kernel_stats.csv; head -8 gpurun_out/${R}_rocprof_kernel_stats.csv | cut -c1-160
# decode: PMC traffic (a warm-up step and a timed one: figures per step = sums / 2)
if [ -z "$SKIP_PMC" ]; then
P="python bench.py --steps 1 --warmup 1 --no-swiftpng --no-cpu-baseline --no-extras"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $P > /dev/null 2> gpurun_out/prof_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $P > /dev/null 2> gpurun_out/prof_write.err
python tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write swiftpng 1024 32 gpurun_out/${R}_pmc_traffic.json 2 > gpurun_out/pmc.log 2>&1; tail -12 gpurun_out/pmc.log; lap "decode stats + PMC"
cp gpurun_out/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json   # (the bench line below reads it: traffic of this very build)
fi
if [ -z "$SKIP_ENC" ]; then
# encode: kernel-trace stats of the encode step, PMC traffic and instruction counters of its deflate
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_enc -- python bench.py --mode encode --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_bench_encode_under_rocprof.json 2> gpurun_out/prof_enc.err
find gpurun_out/prof_enc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_rocprof_encode_kernel_stats.csv; head -6 gpurun_out/${R}_rocprof_encode_kernel_stats.csv | cut -c1-160
PROBE_WHICH=random PROBE_N=1024 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/enc_fetch -- python tools/probe_deflate2.py > /dev/null 2> gpurun_out/enc_fetch.err
PROBE_WHICH=random PROBE_N=1024 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/enc_write -- python tools/probe_deflate2.py > /dev/null 2> gpurun_out/enc_write.err
python tools/pmc_encode.py gpurun_out/enc_fetch gpurun_out/enc_write 1024 gpurun_out/${R}_pmc_encode.json > gpurun_out/pmc_enc.log 2>&1; tail -5 gpurun_out/pmc_enc.log; lap "encode stats + PMC"
cp gpurun_out/${R}_pmc_encode.json profiles/${R}_pmc_encode.json
fi
# (instruction counters of the deflate kernels: profiles/r05k_pmc_l6_insts.json, taken earlier in the round on the level-6 probe)
# the headline line last: it carries the traffic measured above
[ -n "$SKIP_BENCH" ] || { timeout 1500 python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench.err; head -c 1500 gpurun_out/${R}_bench_n1.json; echo; lap "bench"; }
# probes (last: what a budget that runs out may cut)
[ -n "$SKIP_PROBES" ] || { timeout 400 python tools/probe_groups.py --kind zlib --unique 4 > gpurun_out/${R}_probe_groups.log 2>&1; head -3 gpurun_out/${R}_probe_groups.log
PROBE_LEVEL=6 PROBE_WHICH=synth4k,photo PROBE_N=256 timeout 300 python tools/probe_deflate2.py > gpurun_out/${R}_probe_l6_256.log 2>&1; tail -2 gpurun_out/${R}_probe_l6_256.log; }
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_enc gpurun_out/enc_fetch gpurun_out/enc_write gpurun_out/enc_insts
# phase cycle counters of one decode wave / one resolve workgroup (a -DSPNG_D_PROF build of the shipped source)
[ -n "$SKIP_PROBES" ] || [ ! -f variants/libspng_g_prof.so ] || { SPNG_LIB=/root/repo/variants/libspng_g_prof.so timeout 200 python tools/probe_v2.py --kinds swiftpng --steps 1 > gpurun_out/${R}_dprof.log 2>&1; grep -E "^(decode|resolve)" gpurun_out/${R}_dprof.log | sort | uniq -c | sort -rn | head -4 | cut -c1-400; }
