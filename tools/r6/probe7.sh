#!/bin/bash
# Round 6, call 7: weight-major XCD map of the whole-image halo conv + the 8x8 -> 16x16 upsampling conv on that kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== conv op tests"; timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | grep -v amdgpu | tail -6
echo "== shapes A/B (default / UR_HIMG_NOWMAJOR=1 / UR_IGEMM_NOHIMGUPS=1)"
ONLY="@16" timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu
echo "-- nowmajor"
ONLY="@16" UR_HIMG_NOWMAJOR=1 timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu
echo "-- nohimgups"
ONLY="@16up" UR_IGEMM_NOHIMGUPS=1 timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu
echo "== GEMM shapes A/B (default / UR_GEMM_NOWMAJOR=1)"
ONLY="T256" timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu
ONLY="T64" timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu | grep T64
echo "-- nowmajor"
ONLY="T256" UR_GEMM_NOWMAJOR=1 timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu
ONLY="T64" UR_GEMM_NOWMAJOR=1 timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu | grep T64
echo "== forward A/B"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>$O/r6_g_bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])" || tail -5 $O/r6_g_bench_err.txt
UR_GEMM_NOWMAJOR=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no gemm wmajor', d['ms_per_step'])"
UR_GEMM_NOWMAJOR=1 UR_HIMG_NOWMAJOR=1 UR_IGEMM_NOHIMGUPS=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'])"
done
