#!/bin/bash
# Full GPU suite (with durations) + the driver-style bench line; outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O; T=${1:-r6_f}
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/${T}_gpu_tests.log 2>&1
grep -v amdgpu.ids $O/${T}_gpu_tests.log | tail -40
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err
tail -c 1500 $O/${T}_bench.json
