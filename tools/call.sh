cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
PROBE_LEVEL=9 PROBE_WHICH=distinct PROBE_N=1024 timeout 600 python tools/probe_deflate2.py 2>&1 | tail -1 | cut -c1-300
PROBE_LEVEL=9 PROBE_WHICH=random PROBE_N=1024 timeout 400 python tools/probe_deflate2.py 2>&1 | tail -1 | cut -c1-300
SPNG_LIB=/root/repo/variants/libspng_dflprof.so PROBE_LEVEL=9 PROBE_WHICH=distinct PROBE_N=64 timeout 300 python tools/probe_deflate2.py > gpurun_out/r06_dflprof_random.log 2>&1
grep "dfl2_parse prof" gpurun_out/r06_dflprof_random.log | tail -4 | cut -c1-330; tail -1 gpurun_out/r06_dflprof_random.log | cut -c1-250
