#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_chain_gpu.py -x -q 2>&1 | tail -3
ONLY="c3" timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids | grep "@32\|@16\|@8"
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench6.json 2> gpurun_out/bench6.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench6.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','fp16')})
for k,v in sorted(d['families'].items(), key=lambda kv:-kv[1]['ms'])[:8]: print(k, v['launches'], v['ms'], v.get('tflops'))
print(d['parity_vs_oracle']['bf16'], d['parity_vs_oracle']['fp16'])
PY
