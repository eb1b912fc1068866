#!/bin/bash
# end-of-round record (round 6): full GPU suite, default bench, rocprofv3 kernel stats of the bench command (+ the in-graph family table
# bench.py attaches to its line), per-shape event table, phase times, step trace.   tools/r6/final_run.sh <tag>
T=${1:-r6_z}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/${T}_gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu $O/${T}_gpu_tests.log | tail -4
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $T -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-fp16 --no-other-configs > $O/${T}_prof_bench.json 2> $O/${T}_prof.err
f=$(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/summarize_rocprof.py $f $O/${T}_rocprofv3_kernel_stats.csv 200 > /dev/null
python tools/r6/in_graph_families.py $O/${T}_rocprofv3_kernel_stats.csv 5 $O/${T}_in_graph_families.json
cp $O/${T}_in_graph_families.json profiles/r6_in_graph_families.json       # what the bench line below attaches (same box, same tree)
rm -rf $O/prof
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"
timeout 600 python tools/prof_shapes.py 110 > $O/${T}_per_shape_hip_events.txt 2>&1
python tools/phase_times.py 2>&1 | grep -v amdgpu > $O/${T}_phase_times.txt
rm -rf $O/st
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/st -o st -- python tools/step_trace.py > $O/st.log 2>&1
f=$(ls $O/st/*kernel_trace.csv $O/st/*/*kernel_trace.csv 2>/dev/null | head -1)
python tools/step_trace.py --summarize $f > $O/${T}_step_kernel_trace.txt 2>&1
rm -rf $O/st
python - <<PY
import json
d=json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d.get("fp16"), d.get("cpu_baseline"))
print(d.get("parity_vs_oracle",{}).get("bf16"), d.get("parity_vs_oracle",{}).get("fp16"))
print({k:v for k,v in d.get("roofline",{}).items() if k!="traffic_note"}); print(d.get("other_configs"))
print({k:(v["ms"],v["launches"]) for k,v in d["families"].items()})
print(d.get("families_in_graph"))
PY
cat $O/${T}_phase_times.txt; head -12 $O/${T}_rocprofv3_kernel_stats.csv; head -4 $O/${T}_step_kernel_trace.txt
