cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/loop_latency 2>&1 | tail -3
for i in 1 2 3; do timeout -k 10 600 python -m pytest tests/test_gpu_host_mirror.py tests/test_gpu_host_tick.py tests/test_gpu_server.py tests/test_gpu_api.py -x -q 2>&1 | tail -1; done
