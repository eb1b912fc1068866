#!/bin/bash
# Round 5, first GPU call: (1) LDS-DMA vs VGPR-load path probe, (2) per-shape baseline, (3) fresh PMC passes on the three worst
# per-layer GEMM shapes, (4) throttle / violation status under the dominant conv loop (amd-smi), (5) whole-forward baseline.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
echo "== dma_vmem_probe"; timeout 120 tools/probe/dma_vmem_probe > $O/r5_dma_vmem_probe.txt 2>&1; tail -45 $O/r5_dma_vmem_probe.txt
echo "== bench_shapes"; timeout 600 python tools/bench_shapes.py > $O/r5_a_shapes.txt 2>&1; tail -60 $O/r5_a_shapes.txt
echo "== pmc gemm"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $O/r5_pmc_gemm.txt
for shape in "out/proj 1280->1280 T256" "out/proj 640->640 T1024" "out/proj 1280->1280 T64"; do
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
    rm -rf $O/pmc1
    ONLY="$shape" timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc1 -o p -- python tools/bench_shapes.py > $O/pmc1.log 2>&1
    echo "== $shape :: $ctr" >> $O/r5_pmc_gemm.txt
    python - >> $O/r5_pmc_gemm.txt 2>&1 <<PY
import csv,glob,collections
fs=glob.glob("$O/pmc1/*counter_collection.csv")
if not fs:
    print("no counter file (counter unavailable?)"); print(open("$O/pmc1.log").read()[-600:])
else:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:80]+" grid="+r.get("Grid_Size","?"); agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in agg.items():
        if "gemm" in k: print(k.split("::")[-1], {a: round(b/cnt[k][a]) for a,b in v.items()}, "launches", max(cnt[k].values()))
PY
  done
done
rm -rf $O/pmc1 $O/pmc1.log
cat $O/r5_pmc_gemm.txt
echo "== throttle status under the conv loop"
( python tools/loop_conv.py > /dev/null 2>&1 ) &
PID=$!
sleep 8
{ echo "--- amd-smi metric (under a loop of conv3x3 320->320 @64x64, B=8)"; amd-smi metric -g 0 2>&1 | head -150; echo "--- rocm-smi"; rocm-smi --showpower --showclocks --showperflevel 2>&1 | grep -v "^$" | head -30; } > $O/r5_throttle_conv.txt 2>&1
wait $PID
( python tools/loop_attn.py > /dev/null 2>&1 ) &
PID=$!
sleep 8
{ echo "--- amd-smi metric (under a loop of the d=64 attention launch)"; amd-smi metric -g 0 2>&1 | head -150; } > $O/r5_throttle_attn.txt 2>&1
wait $PID
grep -i -n "throttle\|violation\|power\|clk\|clock\|temperature\|hotspot\|limit" $O/r5_throttle_conv.txt | head -80
echo "== forward baseline"
python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --steps 5 --warmup 2 > $O/r5_a_bench.json 2> $O/r5_a_bench.err; tail -c 1500 $O/r5_a_bench.json
