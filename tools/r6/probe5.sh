#!/bin/bash
# Round 6, fifth GPU call: the persistent halo conv - op tests (under a timeout: a barrier mismatch would hang), shape A/B, forward A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== persistent halo op tests"; timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "persistent_halo" 2>&1 | tail -15
echo "== other conv tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | tail -5
echo "== shapes A/B"
for s in "128 128 512" "256 256 256" "512 512 128" "256 256 128"; do
  timeout 120 python tools/r6/time_conv.py $s 2>&1 | grep c3
  UR_HALO_NOPWS=1 timeout 120 python tools/r6/time_conv.py $s 2>&1 | grep c3
done
echo "== forward A/B"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'])"
UR_HALO_NOPWS=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'])"
done
