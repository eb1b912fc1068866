#!/bin/bash
# usage: tools/pmc_dom.sh "<counters>" [ONLY-pattern]  -> per-kernel counter averages while running one conv shape of tools/bench_shapes.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc1
ONLY="${2:-unet c3 320->320@64}" rocprofv3 --kernel-trace --pmc $1 --output-format csv -d gpurun_out/pmc1 -o p -- python tools/bench_shapes.py > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc1/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:90]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k,v in agg.items():
    if "igemm" in k or "gemm_glds" in k: print(k.split("::")[-1], {a: round(b/cnt[k][a]) for a,b in v.items()}, "launches", max(cnt[k].values()))
PY
rm -rf gpurun_out/pmc1
