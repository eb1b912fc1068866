#!/bin/bash
# Round 6: PMC passes PER ATTENTION KERNEL (verdict item 3): the ping-pong self-attention (+ its combine kernel), the 77-key cross-attention
# and the d = 512 VAE attention, each launch shape of the forward in its own run; separate --pmc passes per counter group.
#   tools/r6/pmc_attention.sh > gpurun_out/r6_pmc_attention.txt        (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/attn_case.py <<'PY'
import sys, math; sys.path.insert(0, ".")
import torch
from unirestore_amd import ops
t, heads, d, tk = (int(v) for v in sys.argv[1:5])
B = 8
c = heads * d
q = torch.randn(B, t, 3 * c, device="cuda").to(torch.bfloat16)
ldvt = (tk + 7) // 8 * 8
if tk == t:
    vt = torch.randn(B, c, ldvt, device="cuda").to(torch.bfloat16)
    f = lambda: ops.attention(q, q[:, :, c:], vt, heads, d, t, tk, 1 / math.sqrt(d), ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * ldvt, batch=B)
else:
    k = torch.randn(1, tk, 2 * c, device="cuda").to(torch.bfloat16); vt = torch.zeros(1, c, ldvt, device="cuda", dtype=torch.bfloat16); vt[:, :, :tk] = torch.randn(1, c, tk)
    f = lambda: ops.attention(q, k, vt, heads, d, t, tk, 1 / math.sqrt(d), ldq=3 * c, ldk=2 * c, bs_q=t * 3 * c, bs_k=0, bs_vt=0, batch=B)
for _ in range(6): f()
torch.cuda.synchronize()
PY
for CASE in "4096 5 64 4096" "1024 10 64 1024" "256 20 64 256" "1024 10 64 77" "256 20 64 77" "4096 1 512 4096"; do
  echo "=== B=8 Tq Heads D Tk = $CASE"
  for C in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf gpurun_out/pmc1
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc1 -o p -- python /tmp/attn_case.py $CASE > /dev/null 2>&1
    python - <<PY
import csv,glob,collections,re
fs=glob.glob("gpurun_out/pmc1/*counter_collection.csv")
if not fs: print("no output for: $C")
else:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        m=re.search(r"(attn\w*_kernel)", r["Kernel_Name"])
        if m: k=m.group(1); agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in agg.items(): print(k, {a: round(b/cnt[k][a]) for a,b in v.items()}, "launches", max(cnt[k].values()))
PY
  done
  rm -rf gpurun_out/pmc1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc1 -o p -- python /tmp/attn_case.py $CASE > /dev/null 2>&1
  echo "-- durations (plain kernel trace: Name, Calls, TotalNs, AvgNs)"; grep -h "attn" gpurun_out/pmc1/*kernel_stats.csv | sed 's/(anonymous namespace):://' | cut -c1-160 | head -3
  rm -rf gpurun_out/pmc1
done
