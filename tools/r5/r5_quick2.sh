#!/bin/bash
# op / module tests + one bench with the per-family profile: tools/r5_quick2.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-q}
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_attention_pp_gpu.py tests/test_exports_gpu.py tests/test_chain_gpu.py -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 > gpurun_out/r5_${T}_bench.json 2> gpurun_out/r5_${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_${T}_bench.json").read().strip().splitlines()[-1])
print({k:round(d[k],2) for k in ("value","ms_per_step")})
print({k:(v["ms"],v["launches"],v["avg_us"]) for k,v in d["families"].items()})
PY
