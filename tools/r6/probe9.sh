#!/bin/bash
# Round 6, call 9: cold weights - L2-warm / Infinity-Cache ring / HBM-cold per launch, deeper GEMM rings under cold weights
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== cold weights (default dispatch)"; timeout 600 python tools/cold_weights.py 2>&1 | grep -v amdgpu
echo "== cold weights, UR_IGEMM_COLDDEEP=1 (linear only)"; ONLY=linear UR_IGEMM_COLDDEEP=1 timeout 600 python tools/cold_weights.py 2>&1 | grep -v amdgpu
trace() {   # $1 = tag; env from the caller
  rm -rf $O/st
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/st -o st -- python tools/step_trace.py > $O/st.log 2>&1
  f=$(ls $O/st/*kernel_trace.csv $O/st/*/*kernel_trace.csv 2>/dev/null | head -1)
  python tools/step_trace.py --summarize $f > $O/r6_i_step_trace_$1.txt 2>&1
  head -3 $O/r6_i_step_trace_$1.txt
  rm -rf $O/st
}
echo "== step trace: default"; trace default
echo "== step trace: colddeep"; UR_IGEMM_COLDDEEP=1 trace colddeep
echo "== forward A/B"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>$O/r6_i_bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'])" || tail -5 $O/r6_i_bench_err.txt
UR_IGEMM_COLDDEEP=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('colddeep', d['ms_per_step'])"
done
