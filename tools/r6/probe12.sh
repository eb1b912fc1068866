#!/bin/bash
# Round 6, call 12: is pick_xcd_grid's choice the best one in-graph?  Step traces with the row-band count forced (UR_XCD_FORCE = 8 / 4 / 2 / 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
trace() {
  rm -rf $O/st
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/st -o st -- python tools/step_trace.py > $O/st.log 2>&1
  f=$(ls $O/st/*kernel_trace.csv $O/st/*/*kernel_trace.csv 2>/dev/null | head -1)
  python tools/step_trace.py --summarize $f > $O/r6_m_step_trace_$1.txt 2>&1
  head -1 $O/r6_m_step_trace_$1.txt
  rm -rf $O/st
}
echo "== chosen"; trace chosen
for g in 8 4 2 1; do echo "== forced $g"; UR_XCD_FORCE=$g trace f$g; done
