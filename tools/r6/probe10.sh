#!/bin/bash
# Round 6, call 10: weight prefetch branch (ur_prefetch on a forked stream) - module tests, forward A/B, step trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== module tests"; timeout 900 python -m pytest tests/test_modules_gpu.py -x -q -k "full_forward or deterministic" 2>&1 | grep -v amdgpu | tail -4
echo "== forward A/B"
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>$O/r6_j_bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch', d['ms_per_step'], d['output_finite'])" || tail -5 $O/r6_j_bench_err.txt
UR_PREFETCH_CHAINS=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch + chains', d['ms_per_step'])"
UR_PREFETCH=0 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no prefetch', d['ms_per_step'])"
done
trace() {   # $1 = tag; env from the caller
  rm -rf $O/st
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/st -o st -- python tools/step_trace.py > $O/st.log 2>&1
  f=$(ls $O/st/*kernel_trace.csv $O/st/*/*kernel_trace.csv 2>/dev/null | head -1)
  python tools/step_trace.py --summarize $f > $O/r6_j_step_trace_$1.txt 2>&1
  head -3 $O/r6_j_step_trace_$1.txt; tail -3 $O/st.log
  rm -rf $O/st
}
echo "== step trace: prefetch"; trace prefetch
echo "== step trace: no prefetch"; UR_PREFETCH=0 trace noprefetch
