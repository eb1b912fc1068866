#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== persistent halo op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "persistent_halo" > $O/r6_e_pws_tests.log 2>&1; grep -v amdgpu.ids $O/r6_e_pws_tests.log | grep -v "^  File\|^Extension" | tail -25
echo "== timelines"
for s in "128 128 512" "256 256 256"; do UR_LIB=unirestore_amd/ab/libur_tl.so timeout 120 python tools/r6/pws_timeline.py $s 2>&1 | grep -v amdgpu; done
echo "== shapes A/B"
for s in "128 128 512" "256 256 256" "512 512 128" "256 256 128"; do
  timeout 120 python tools/r6/time_conv.py $s 2>&1 | grep c3
  UR_HALO_NOPWS=1 timeout 120 python tools/r6/time_conv.py $s 2>&1 | grep c3
done
echo "== forward A/B"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>$O/r6_e_bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])" || tail -5 $O/r6_e_bench_err.txt
UR_HALO_NOPWS=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'])"
done
