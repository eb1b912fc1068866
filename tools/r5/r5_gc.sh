#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_modules_gpu.py -q -k "cfrm or tfa or spade or autoencoder or full_forward" 2>&1 | tail -3
UR_FUSE_LN2D=0 python tools/phase_times.py 2>&1 | grep -E "encode|sum"
python tools/phase_times.py 2>&1 | grep -E "encode|sum"
