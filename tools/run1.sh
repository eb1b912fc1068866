#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
UR_HALO_2X5=1 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | tail -2
ONLY="unet c3" timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids | grep "@64"
ONLY="unet c3" UR_HALO_2X5=1 timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids | grep "@64"
for v in w8 w4 w8 w4; do
  if [ $v = w4 ]; then export UR_HALO_2X5=1; else unset UR_HALO_2X5; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16 > gpurun_out/bench_ab_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/bench_ab_$v.json') if l.startswith('{')][-1])
f=d['families']
print('$v', round(d['ms_per_step'],1), 'gemm', f['gemm1x1_igemm']['ms'], 'conv', f['conv3x3_igemm']['ms'], 'attn', f['attention']['ms'])
PY
done
