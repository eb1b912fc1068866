#!/bin/bash
# Round 6, call 16: fused-GroupNorm thresholds again, now that the halo convs have two loader waves
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for cfg in "65536 256" "16384 512" "4096 512" "16384 256"; do
  set -- $cfg
  UR_FUSE_GN_MIN_PIXELS=$1 UR_FUSE_GN_MAX_COUT=$2 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_pixels $1 max_cout $2:', d['ms_per_step'], d['output_finite'])"
done
done
