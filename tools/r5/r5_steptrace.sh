#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/st
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st -o st -- python tools/step_trace.py > gpurun_out/st.log 2>&1
f=$(ls gpurun_out/st/*kernel_trace.csv gpurun_out/st/*/*kernel_trace.csv 2>/dev/null | head -1)
echo "trace: $f"; head -2 $f | cut -c1-400
python tools/step_trace.py --summarize $f > gpurun_out/r5_step_kernel_trace.txt 2>&1
head -95 gpurun_out/r5_step_kernel_trace.txt
rm -rf gpurun_out/st
