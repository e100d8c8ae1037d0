cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/loop_latency 2>&1 | tail -3
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/s5_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -4 gpurun_out/s5_gpu.log
