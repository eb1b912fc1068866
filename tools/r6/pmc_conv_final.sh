#!/bin/bash
# Round 6, final tree: PMC passes of the conv classes whose kernel changed after the first pass (two loader waves / wave-specialised whole-image
# kernel / weight stream / XCD maps) -> gpurun_out/r6_pmc_conv_final_raw.txt (same format as r6_pmc_conv_raw.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
: > $O/r6_pmc_conv_final_raw.txt
for shape in "vae c3 128->128@512" "vae c3 256->256@256" "unet c3 1280->1280@16" "unet c3 1280->1280@8" "unet c3 320->320@64" "unet c3 1280->1280@16up"; do
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
    rm -rf $O/pmc1
    REPS=4 ONLY="$shape" timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc1 -o p -- python tools/bench_shapes.py > $O/pmc1.log 2>&1
    echo "== $shape :: $ctr" >> $O/r6_pmc_conv_final_raw.txt
    python - >> $O/r6_pmc_conv_final_raw.txt 2>&1 <<PY
import csv,glob,collections
fs=glob.glob("$O/pmc1/*counter_collection.csv")
if not fs:
    print("no counter file (counter unavailable?)"); print(open("$O/pmc1.log").read()[-600:])
else:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:90]+" grid="+r.get("Grid_Size","?")+" wg="+r.get("Workgroup_Size","?"); agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in agg.items():
        if "gemm" in k or "splitk" in k or "wstream" in k: print(k.split("::")[-1], {a: round(b/cnt[k][a]) for a,b in v.items()}, "launches", max(cnt[k].values()))
PY
  done
done
rm -rf $O/pmc1
grep -c "launches" $O/r6_pmc_conv_final_raw.txt
