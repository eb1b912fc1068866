#!/bin/bash
# Round 6, call 14: second loader wave in the 8x32 halo conv (UR_HALO_LD2=1) + the two-loader whole-image kernel (UR_HIMG_WS=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== conv op tests with UR_HALO_LD2=1 UR_HIMG_WS=1"; UR_HALO_LD2=1 UR_HIMG_WS=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | grep -v amdgpu | tail -4
echo "== shapes A/B"
for sel in "unet c3" "vae c3" "ctrlB"; do ONLY="$sel" timeout 300 python tools/bench_shapes.py 2>&1 | grep "c3"; done > $O/r6_o_shapes_default.txt
for sel in "unet c3" "vae c3" "ctrlB"; do ONLY="$sel" UR_HALO_LD2=1 UR_HIMG_WS=1 timeout 300 python tools/bench_shapes.py 2>&1 | grep "c3"; done > $O/r6_o_shapes_ld2.txt
paste -d'|' <(cut -c1-62 $O/r6_o_shapes_default.txt) <(cut -c42-62 $O/r6_o_shapes_ld2.txt)
echo "== forward A/B"
for i in 1 2; do
UR_HALO_LD2=1 UR_HIMG_WS=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ld2+himg_ws', d['ms_per_step'], d['output_finite'])"
UR_HIMG_WS=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('himg_ws only', d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'])"
done
