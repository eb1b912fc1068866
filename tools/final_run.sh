#!/bin/bash
# end-of-round record: full GPU suite, default bench, rocprofv3 kernel stats of the bench command, per-shape event table, step trace.  tools/final_run.sh <tag>
T=${1:-r5_f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s > gpurun_out/${T}_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${T}_gpu_tests.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $T -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-fp16 --no-other-configs > gpurun_out/${T}_prof_bench.json 2> gpurun_out/${T}_prof.err
f=$(ls gpurun_out/prof/*kernel_stats.csv gpurun_out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/summarize_rocprof.py $f gpurun_out/${T}_rocprofv3_kernel_stats.csv 45 > /dev/null
rm -rf gpurun_out/prof
timeout 600 python tools/prof_shapes.py 110 > gpurun_out/${T}_per_shape_hip_events.txt 2>&1
python tools/phase_times.py 2>&1 | grep -v amdgpu > gpurun_out/${T}_phase_times.txt
bash tools/r5/r5_steptrace.sh > /dev/null 2>&1; cp gpurun_out/r5_step_kernel_trace.txt gpurun_out/${T}_step_kernel_trace.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d.get("fp16"), d.get("cpu_baseline"))
print(d.get("parity_vs_oracle",{}).get("bf16"), d.get("parity_vs_oracle",{}).get("fp16"))
print(d.get("roofline")); print(d.get("other_configs"))
print({k:(v["ms"],v["launches"]) for k,v in d["families"].items()})
PY
cat gpurun_out/${T}_phase_times.txt; head -12 gpurun_out/${T}_rocprofv3_kernel_stats.csv
