cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_dist_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -8
