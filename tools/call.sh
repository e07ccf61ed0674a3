cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06q_pytest_gpu.log 2>&1; tail -5 gpurun_out/r06q_pytest_gpu.log; echo "[t+$(( $(date +%s) - T0 )) s]"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
