#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/fp16_zt_attrib.py default 2>&1 | grep "^\["
UR_CHAIN=0 python tools/fp16_zt_attrib.py "UR_CHAIN=0 (per-layer transformer at 64x64)" 2>&1 | grep "^\["
UR_CHAIN=0 UR_FUSE_LN=0 python tools/fp16_zt_attrib.py "UR_CHAIN=0 UR_FUSE_LN=0 (separate LayerNorm passes)" 2>&1 | grep "^\["
UR_ATTN_NOPP=1 python tools/fp16_zt_attrib.py "UR_ATTN_NOPP=1 (round-1 attention kernel)" 2>&1 | grep "^\["
UR_FUSE_GN=0 python tools/fp16_zt_attrib.py "UR_FUSE_GN=0" 2>&1 | grep "^\["
