#!/bin/bash
# same-box A/B of two builds of the library: microbench + whole forward, interleaved twice
cd $GRAFT_REPO_ROOT
OLD=unirestore_amd/ab/lib_head.so
echo "== old"; UR_LIB=$OLD python tools/bench_fin.py 2>&1 | grep -v amdgpu
echo "== new"; python tools/bench_fin.py 2>&1 | grep -v amdgpu
for rep in 1 2; do
  for lib in $OLD ""; do
    ms=$(UR_LIB=$lib python bench.py --no-cpu-baseline --no-profile --no-other-configs --no-fp16 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "lib=${lib:-new} : $ms ms"
  done
done
bash tools/r5_attrib.sh
