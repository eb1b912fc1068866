#!/bin/bash
# full GPU suite + default bench of the current tree: tools/r5_run_tests.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-x}
timeout 2400 python -m pytest tests -m gpu -x -q -s > gpurun_out/r5_${T}_tests.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r5_${T}_tests.log
grep -E "heavy tail|configs\[1\] sample|per-sample" gpurun_out/r5_${T}_tests.log | head -20
python bench.py > gpurun_out/r5_${T}_bench.json 2> gpurun_out/r5_${T}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_${T}_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d.get("fp16"), d.get("cpu_baseline"), d.get("parity_vs_oracle",{}).get("bf16"), d.get("parity_vs_oracle",{}).get("fp16"))
print({k:(v["ms"],v["launches"]) for k,v in d["families"].items()})
PY
