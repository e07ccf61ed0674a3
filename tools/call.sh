cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_groups.py --kind swiftpng --unique 4 --shapes 1x1,8x1,32x1,128x1,128x2 > gpurun_out/r06s_probe_groups_swiftpng.log 2>&1; grep "call(s)" gpurun_out/r06s_probe_groups_swiftpng.log | cut -c1-230
