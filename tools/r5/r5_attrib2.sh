#!/bin/bash
cd $GRAFT_REPO_ROOT
for spec in "UR_X=1" "UR_IGEMM_NOHALO=1" "UR_IGEMM_NOHIMG=1" "UR_IGEMM_NOHSPLIT=1" "UR_HALO_NOWS=1" "UR_IGEMM_NOGNRED=1" "UR_IGEMM_NOG1DMA=1" "UR_KCM=0"; do
  env $spec python tools/fp16_zt_attrib.py "$spec" 2>&1 | grep "^\[" | grep fp16
done
