#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "thin or conv" 2>&1 | tail -3
for v in 1 0; do
  if [ $v = 1 ]; then export UR_IGEMM_NOTHIN=1; else unset UR_IGEMM_NOTHIN; fi
  echo "== UR_IGEMM_NOTHIN=${UR_IGEMM_NOTHIN:-unset}"
  python tools/phase_times.py 2>&1 | grep -E "decode \+|step|encode \+"
done
