#!/bin/bash
# Shader clock and socket power while a workload runs: tools/power_probe.sh "<python command>"   (run on the GPU box)
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -E "sclk|Power|perf" | head -6
echo "--- under load"
( $1 > /dev/null 2>&1 ) &
PID=$!
sleep ${2:-6}
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics|Package Power" | tr '\n' ' '; echo; sleep 0.5; done
wait $PID
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -2
