#!/bin/bash
# round 4, first GPU call: access-pattern calibration, unfilter variants, the new parity tests
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 ./variants/probe_copy 128 > gpurun_out/r04_probe_copy.log 2>&1; tail -3 gpurun_out/r04_probe_copy.log
for v in base cs base_nc cs_nc; do
  SPNG_LIB=$PWD/variants/libspng_$v.so timeout 300 python tools/probe_unfilter_variants.py >> gpurun_out/r04_probe_unfilter_variants.log 2>&1
done
for v in base_a16400 cs_a16400; do
  PROBE_STRIDE=16400 SPNG_LIB=$PWD/variants/libspng_$v.so timeout 300 python tools/probe_unfilter_variants.py >> gpurun_out/r04_probe_unfilter_variants.log 2>&1
done
for v in cs_a16512 cs_a16512_nc; do
  PROBE_STRIDE=16512 SPNG_LIB=$PWD/variants/libspng_$v.so timeout 300 python tools/probe_unfilter_variants.py >> gpurun_out/r04_probe_unfilter_variants.log 2>&1
done
cat gpurun_out/r04_probe_unfilter_variants.log
timeout 900 python -m pytest tests -m gpu -x -q -k "unfilter or vertex_cap or whole_stream or single_push or retry_pass or config5 or pngsuite or resume" > gpurun_out/r04_pytest_subset.log 2>&1; tail -5 gpurun_out/r04_pytest_subset.log
