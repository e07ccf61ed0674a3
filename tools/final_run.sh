#!/bin/bash
# Round-end measurement on the GPU box, ONE gpurun call: parity suite, smoke, kernel-trace stats of the decode step and of the encode
# step, the PMC passes of the level-9 deflate (FETCH_SIZE / WRITE_SIZE in separate passes), then the headline bench line with
# `--traffic` (its own PMC sub-runs of the decode step, taken inside the invocation), the driver's own command, phase counters.
# Everything lands in gpurun_out/ (the files quoted in DESIGN.md / profiles/README.md are copied to profiles/ afterwards).
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
export R=${R:-r06}
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $1"; }
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${R}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${R}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; tail -1 gpurun_out/${R}_smoke.log; lap "tests + smoke"
# decode: kernel-trace stats of the bench command (1 warm-up + 3 timed steps of the headline workload)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python /root/repo/bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline --no-extras > /root/repo/gpurun_out/${R}_bench_under_rocprof.json 2> /tmp/prof_stats.err)
find /tmp/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_rocprof_kernel_stats.csv; head -6 gpurun_out/${R}_rocprof_kernel_stats.csv | cut -c1-160; lap "decode kernel stats"
if [ -z "$SKIP_ENC" ]; then
# encode: kernel-trace stats of the encode step, PMC traffic of its deflate (a warm-up call and a timed one: figures per call)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -- python /root/repo/bench.py --mode encode --steps 1 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/${R}_bench_encode_under_rocprof.json 2> /tmp/prof_enc.err)
find /tmp/prof_enc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${R}_rocprof_encode_kernel_stats.csv; head -5 gpurun_out/${R}_rocprof_encode_kernel_stats.csv | cut -c1-160
(cd /tmp && PROBE_WHICH=random PROBE_N=1024 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/enc_fetch -- python /root/repo/tools/probe_deflate2.py > /dev/null 2> /tmp/enc_fetch.err)
(cd /tmp && PROBE_WHICH=random PROBE_N=1024 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/enc_write -- python /root/repo/tools/probe_deflate2.py > /dev/null 2> /tmp/enc_write.err)
python tools/pmc_encode.py /tmp/enc_fetch /tmp/enc_write 1024 gpurun_out/${R}_pmc_encode.json > gpurun_out/pmc_enc.log 2>&1; tail -3 gpurun_out/pmc_enc.log; lap "encode stats + PMC"
cp gpurun_out/${R}_pmc_encode.json profiles/${R}_pmc_encode.json   # (the bench line below reads it: traffic of this very build)
fi
# the headline line: `--traffic` takes the decode step's PMC passes inside the invocation (profiles/r06_pmc_traffic.json is rewritten)
[ -n "$SKIP_BENCH" ] || { timeout 2400 python bench.py --traffic > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench.err; head -c 1200 gpurun_out/${R}_bench_n1.json; echo; tail -2 gpurun_out/${R}_bench.err; cp profiles/${R}_pmc_traffic.json gpurun_out/ 2>/dev/null; lap "bench --traffic"; }
# the driver's own command
[ -n "$SKIP_DRIVER" ] || { timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_cmd.json 2> gpurun_out/${R}_bench_driver.err; head -c 400 gpurun_out/${R}_bench_driver_cmd.json; echo; lap "driver's command"; }
# phase cycle counters of one decode wave / one resolve workgroup (a -DSPNG_D_PROF build of the shipped source)
[ ! -f variants/libspng_prof.so ] || { SPNG_LIB=/root/repo/variants/libspng_prof.so timeout 200 python tools/probe_v2.py --kinds swiftpng --steps 1 > gpurun_out/${R}_dprof.log 2>&1; grep -E "^(decode|resolve)" gpurun_out/${R}_dprof.log | sort | uniq -c | sort -rn | head -3 | cut -c1-400; }
rm -rf /tmp/prof_stats /tmp/prof_enc /tmp/enc_fetch /tmp/enc_write
lap "done"
