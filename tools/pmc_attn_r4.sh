#!/bin/bash
# round 4: PMC passes + socket power / clock of the d=64 self-attention launch with the ping-pong kernel and (UR_ATTN_NOPP=1) the round-1 kernel
cd $GRAFT_REPO_ROOT
for nopp in 0 1; do
  echo "=== UR_ATTN_NOPP=$nopp"
  UR_ATTN_NOPP=$nopp bash tools/pmc_attn.sh 2>&1 | tail -7
  echo "--- power / clock under a 12 s loop of the launch"
  UR_ATTN_NOPP=$nopp bash tools/power_probe.sh "python tools/loop_attn.py" 5 2>&1 | tail -9
done
