cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); lap() { echo "[t+$(( $(date +%s) - T0 )) s] $1"; }
timeout 600 python -m pytest tests/test_gpu_decode.py -q -k "different_lengths or deflate_vs_oracle or encode_batch_end_to_end or block_boundaries" -x 2>&1 | tail -3; lap tests
timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline --legs small_images,shard_probe,single_image > gpurun_out/r06a_bench_legs.json 2> gpurun_out/r06a_bench_legs.err; tail -c 3000 gpurun_out/r06a_bench_legs.json; tail -5 gpurun_out/r06a_bench_legs.err; lap bench
