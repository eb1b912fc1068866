#!/bin/bash
# same-box A/B of the committed library (unirestore_amd/ab/lib_head.so, built by the caller) against the working tree, interleaved
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_chain_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in unirestore_amd/ab/lib_head.so ""; do
    ms=$(UR_LIB=$lib python bench.py --no-cpu-baseline --no-profile --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "lib=${lib:-new} : $ms ms"
  done
done
