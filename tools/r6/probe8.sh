#!/bin/bash
# Round 6, call 8: 2-D XCD partition of the GEMM tile grids (pick_xcd_grid) - op tests, in-graph step traces new / old, forward A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== op tests (linear / conv / chain / modules)"; timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | grep -v amdgpu | tail -4
trace() {   # $1 = tag; env from the caller
  rm -rf $O/st
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/st -o st -- python tools/step_trace.py > $O/st.log 2>&1
  f=$(ls $O/st/*kernel_trace.csv $O/st/*/*kernel_trace.csv 2>/dev/null | head -1)
  python tools/step_trace.py --summarize $f > $O/r6_h_step_trace_$1.txt 2>&1
  head -3 $O/r6_h_step_trace_$1.txt
  rm -rf $O/st
}
echo "== step trace: new"; trace new
echo "== step trace: old"; UR_NOXCDGRID=1 UR_HIMG_NOWMAJOR=1 UR_IGEMM_NOHIMGUPS=1 trace old
echo "== step trace: xcd grid only (no himg wmajor)"; UR_HIMG_NOWMAJOR=1 trace nohimgw
echo "== forward A/B"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>$O/r6_h_bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])" || tail -5 $O/r6_h_bench_err.txt
UR_NOXCDGRID=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no xcd grid', d['ms_per_step'])"
UR_NOXCDGRID=1 UR_HIMG_NOWMAJOR=1 UR_IGEMM_NOHIMGUPS=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'])"
done
