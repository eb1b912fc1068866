#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench5.json 2> gpurun_out/bench5.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench5.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','fp16')})
for k,v in sorted(d['families'].items(), key=lambda kv:-kv[1]['ms']): print(k, v['launches'], v['ms'], v.get('tflops'))
print(d['parity_vs_oracle'])
PY
