cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
P="python bench.py --steps 1 --warmup 0 --no-swiftpng --no-cpu-baseline"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $P > /dev/null 2> gpurun_out/prof_fetch.err
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $P > /dev/null 2> gpurun_out/prof_write.err
rm -f gpurun_out/r02_pmc_traffic.json
python tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write zlib 1024 32 gpurun_out/r02_pmc_traffic.json > gpurun_out/pmc.log 2>&1; grep -A6 pinf_emit gpurun_out/pmc.log
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
