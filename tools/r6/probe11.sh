#!/bin/bash
# Round 6, call 11: address-translation probe + packed weights in one arena (UR_WEIGHT_ARENA)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== tlb probe"; timeout 300 python tools/r6/tlb_probe.py 2>&1 | grep -v amdgpu
echo "== forward A/B: arena"
for i in 1 2; do
UR_WEIGHT_ARENA=3 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>$O/r6_l_bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('arena', d['ms_per_step'], d['output_finite'])" || tail -5 $O/r6_l_bench_err.txt
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'])"
done
