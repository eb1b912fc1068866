#!/bin/bash
# same-box A/B of the whole forward under environment switches: tools/ab_env.sh "UR_POST_GN=0" "UR_POST_GN=1" ...  (each twice, interleaved)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spec in "$@"; do
    ms=$(env $spec python bench.py --no-cpu-baseline --no-profile --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$spec : $ms ms"
  done
done
