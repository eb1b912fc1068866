#!/bin/bash
# Round 6: HBM-side traffic of the weight-heavy launches with the XCD grids on / off (FETCH_SIZE, WRITE_SIZE: separate passes).
#   tools/r6/pmc_xcd.sh > gpurun_out/r6_pmc_xcd_raw.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/xcd_case.py <<'PY'
import sys, math; sys.path.insert(0, ".")
import torch
from unirestore_amd import ops
kind = sys.argv[1]
if kind == "geglu16":      # 2048 x 1280 -> 2 x 5120 (the 16x16-level GEGLU)
    pc = ops.pack_conv(torch.randn(10240, 1280) / 1280 ** 0.5, torch.randn(10240), "cuda", pair=True)
    x = torch.randn(2048, 1280, device="cuda").to(torch.bfloat16)
    f = lambda: ops.linear(x, pc, act=ops.UR_ACT_GEGLU)
elif kind == "qkv16":      # 2048 x 1280 -> 3840
    pc = ops.pack_conv(torch.randn(3840, 1280) / 1280 ** 0.5, None, "cuda")
    x = torch.randn(2048, 1280, device="cuda").to(torch.bfloat16)
    f = lambda: ops.linear(x, pc)
elif kind == "lin8":       # 512 x 1280 -> 1280 (+ residual)
    pc = ops.pack_conv(torch.randn(1280, 1280) / 1280 ** 0.5, torch.randn(1280), "cuda")
    x = torch.randn(512, 1280, device="cuda").to(torch.bfloat16); r = torch.randn(512, 1280, device="cuda").to(torch.bfloat16)
    f = lambda: ops.linear(x, pc, residual=r)
elif kind == "geglu8":     # 512 x 1280 -> 2 x 5120
    pc = ops.pack_conv(torch.randn(10240, 1280) / 1280 ** 0.5, torch.randn(10240), "cuda", pair=True)
    x = torch.randn(512, 1280, device="cuda").to(torch.bfloat16)
    f = lambda: ops.linear(x, pc, act=ops.UR_ACT_GEGLU)
elif kind == "conv16":     # 3x3 1280 -> 1280 @16x16, B = 8
    pc = ops.pack_conv(torch.randn(1280, 1280, 3, 3) / (1280 * 9) ** 0.5, torch.randn(1280), "cuda")
    x = torch.randn(8, 16, 16, 1280, device="cuda").to(torch.bfloat16)
    f = lambda: ops.conv(x, pc, gn=True)
elif kind == "conv16up":   # 3x3 1280 -> 1280, 8x8 -> 16x16
    pc = ops.pack_conv(torch.randn(1280, 1280, 3, 3) / (1280 * 9) ** 0.5, torch.randn(1280), "cuda")
    x = torch.randn(8, 8, 8, 1280, device="cuda").to(torch.bfloat16)
    f = lambda: ops.conv(x, pc, upsample=True, gn=True)
for _ in range(8): f()
torch.cuda.synchronize()
PY
for CASE in geglu16 qkv16 lin8 geglu8 conv16 conv16up; do
  for MODE in grid old; do
    if [ $MODE = old ]; then export UR_NOXCDGRID=1 UR_HIMG_NOWMAJOR=1 UR_IGEMM_NOHIMGUPS=1; else unset UR_NOXCDGRID UR_HIMG_NOWMAJOR UR_IGEMM_NOHIMGUPS; fi
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf gpurun_out/pmc1
      timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc1 -o p -- python /tmp/xcd_case.py $CASE > /dev/null 2>&1
      python - <<PY
import csv,glob,collections,re
fs=glob.glob("gpurun_out/pmc1/*counter_collection.csv")
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(fs[0])) if fs else []:
    n=r["Kernel_Name"]
    if "igemm" in n or "gemm_glds" in n or "splitk" in n or "ConvK" in n:
        k=re.sub(r"\(anonymous namespace\)::|^void ","",n)[:60]; agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
print("$CASE $MODE $C", {k: round(v/cnt[k]) for k,v in agg.items()}, {k: cnt[k] for k in cnt})
PY
    done
  done
done
rm -rf gpurun_out/pmc1
