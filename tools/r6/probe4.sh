#!/bin/bash
# Round 6, fourth GPU call: wstream v2 (patch-first prologue, two chunks per wave at K = 9 x 2560): tests, A/B, is it live in the model?
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "weight_stream or feature_matrix_small_batch" 2>&1 | tail -3
echo "== shapes A/B (8x8)"
ONLY="@8" timeout 300 python tools/bench_shapes.py 2>&1 | grep "c3"
UR_IGEMM_NOWSTREAM=1 ONLY="@8" timeout 300 python tools/bench_shapes.py 2>&1 | grep "c3"
echo "== per-shape events of the forward: M512 convs, new vs old"
timeout 600 python tools/prof_shapes.py 400 2>&1 | grep "total profiled\|c3 M512" 
UR_IGEMM_NOWSTREAM=1 timeout 600 python tools/prof_shapes.py 400 2>&1 | grep "total profiled\|c3 M512"
echo "== forward A/B"
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])"
UR_IGEMM_NOWSTREAM=1 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'])"
done
