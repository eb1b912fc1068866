#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "v_transposed or grouped_colsum or layernorm_fused" 2>&1 | tail -3
python tools/fp16_zt_attrib.py "after the V^T fix" 2>&1 | grep "^\["
timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -s -k "config1_sample" 2>&1 | grep -E "rel-L2|passed|failed"
