cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_v2.py --kinds swiftpng --steps 3 > gpurun_out/r06f_probe_big_tiles.log 2>&1; grep -E "^(swiftpng|zlib) auto" gpurun_out/r06f_probe_big_tiles.log | cut -c1-200
SPNG_EXP_SMALL_TILES=1 timeout 600 python tools/probe_v2.py --kinds swiftpng,zlib --steps 3 > gpurun_out/r06f_probe_small_tiles.log 2>&1; grep -E "^(swiftpng|zlib) auto" gpurun_out/r06f_probe_small_tiles.log | cut -c1-200
