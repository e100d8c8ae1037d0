cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_logdist.py -x -q 2>&1 | tail -3
