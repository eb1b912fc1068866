#!/bin/bash
# PMC passes over the d=64 self-attention launch (B=8, T=4096, 5 heads): tools/pmc_attn.sh   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/attn_run.py <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from unirestore_amd import ops
B, t, heads, d = 8, 4096, 5, 64
c = heads * d
qkv = torch.randn(B, t, 3 * c, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, c, t, device="cuda").to(torch.bfloat16)
for _ in range(6):
    ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, 0.125, ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=B)
torch.cuda.synchronize()
PY
for C in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
rm -rf gpurun_out/pmc1
rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc1 -o p -- python /tmp/attn_run.py > /dev/null 2>&1
python - <<PY
import csv,glob,collections
fs=glob.glob("gpurun_out/pmc1/*counter_collection.csv")
if not fs: print("no output for: $C")
else:
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "attn_" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    print({a: round(b/cnt[a]) for a,b in agg.items()}, "launches", max(cnt.values()) if cnt else 0)
PY
done
rm -rf gpurun_out/pmc1
# kernel duration from a plain kernel trace (un-profiled clocks differ: compare like with like)
rm -rf gpurun_out/pmc1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc1 -o p -- python /tmp/attn_run.py > /dev/null 2>&1
grep -h "attn_" gpurun_out/pmc1/*kernel_stats.csv | head -3
rm -rf gpurun_out/pmc1
