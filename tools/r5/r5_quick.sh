#!/bin/bash
# quick check of a kernel-level change: op tests + shape table + short forward A/B under env switches given as arguments
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_attention_pp_gpu.py tests/test_exports_gpu.py -x -q 2>&1 | tail -3
for spec in "$@"; do
  ms=$(env $spec python bench.py --no-cpu-baseline --no-profile --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$spec : $ms ms"
done
