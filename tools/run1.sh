#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for v in asm noasm asm noasm; do
  if [ $v = noasm ]; then export UR_LIB=$PWD/unirestore_amd/ab/libur_gemm_noasm.so; else unset UR_LIB; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16 > gpurun_out/bench_ab_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/bench_ab_$v.json') if l.startswith('{')][-1])
f=d['families']
print('$v', round(d['ms_per_step'],1), 'gemm', f['gemm1x1_igemm']['ms'], 'conv', f['conv3x3_igemm']['ms'], 'attn', f['attention']['ms'])
PY
done
