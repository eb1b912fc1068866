#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_r3_d.json 2> gpurun_out/bench_r3_d.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r3_d.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','fp16','steps','warmup')})
print(d['roofline'])
for k,v in sorted(d['families'].items(), key=lambda kv:-kv[1]['ms'])[:9]: print(k, v['launches'], v['ms'], v.get('tflops'))
print(d['cpu_baseline']); print(d['parity_vs_oracle']['bf16'], d['parity_vs_oracle']['fp16'])
PY
