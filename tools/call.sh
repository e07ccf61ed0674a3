cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
SPNG_LIB=/root/repo/variants/libspng_dflprof.so PROBE_LEVEL=9 PROBE_WHICH=bench_photo PROBE_N=8 timeout 300 python tools/probe_deflate2.py > gpurun_out/r06_dflprof_bphoto.log 2>&1
grep -c "dfl2_parse prof" gpurun_out/r06_dflprof_bphoto.log; grep "dfl2_parse prof" gpurun_out/r06_dflprof_bphoto.log | tail -12 | cut -c1-330
