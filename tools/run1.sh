cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_exports_gpu.py tests/test_checkpoint_gpu.py tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -15
