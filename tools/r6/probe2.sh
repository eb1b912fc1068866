#!/bin/bash
# Round 6, second GPU call: the weight-streaming 8x8 conv - op tests, per-shape A/B against the tiled kernel, kernel trace, forward A/B.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "weight_stream or feature_matrix_small_batch" 2>&1 | tail -5
echo "== shapes A/B (8x8)"
ONLY="@8" timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/r6_b_shapes8_new.txt
UR_IGEMM_NOWSTREAM=1 ONLY="@8" timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/r6_b_shapes8_old.txt
echo "== kernel trace"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/kt1
REPS=4 ONLY="unet c3 1280->1280@8" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o p -- python tools/bench_shapes.py > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$O/kt1/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]: print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
rm -rf $O/kt1
echo "== forward A/B"
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])"
UR_IGEMM_NOWSTREAM=1 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'])"
done
