#!/bin/bash
# Round 6, call 13: wave-specialised whole-image halo conv (UR_HIMG_WS=1) - op tests under a timeout (a barrier mismatch would hang), bit identity, A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== conv op tests with UR_HIMG_WS=1"; UR_HIMG_WS=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | grep -v amdgpu | tail -4
echo "== bit identity ws vs self-loading"
cat > /tmp/ws_case.py <<'PY'
import sys, math, torch; sys.path.insert(0, ".")
from unirestore_amd import ops
outs = []
for dt in ("bf16", "fp16"):
    ops.set_dtype(dt); DT = ops.act_dtype()
    for (n, cin, cout, hw, ups, c2) in [(8, 1280, 1280, 16, False, 0), (8, 1280, 1280, 16, False, 1280), (3, 256, 384, 16, False, 0), (8, 512, 1280, 8, True, 0), (8, 640, 1280, 16, False, 0), (4, 512, 512, 8, False, 0)]:
        g = torch.Generator().manual_seed(cin + cout + hw)
        x = torch.randn(n, hw, hw, cin, generator=g).to(DT).cuda()
        x2 = torch.randn(n, hw, hw, c2, generator=g).to(DT).cuda() if c2 else None
        wt = (torch.randn(cout, cin + c2, 3, 3, generator=g) / math.sqrt((cin + c2) * 9)).to(DT).float()
        r = torch.randn(n, hw * (2 if ups else 1), hw * (2 if ups else 1), cout, generator=g).to(DT).cuda()
        y = ops.conv(x, ops.pack_conv(wt, torch.randn(cout, generator=g), "cuda", c1=cin if c2 else None), x2=x2, upsample=ups, residual=r, gn=True)
        outs.append(y.cpu()); outs.append(ops.gn_of(y)[0].cpu())
torch.save(outs, sys.argv[1])
PY
UR_HIMG_WS=1 timeout 300 python /tmp/ws_case.py /tmp/ws.pt 2>&1 | grep -v amdgpu; timeout 300 python /tmp/ws_case.py /tmp/base.pt 2>&1 | grep -v amdgpu
python -c "
import torch
a=torch.load('/tmp/ws.pt'); b=torch.load('/tmp/base.pt')
print('tensors', len(a), 'all bit-identical:', all(torch.equal(x,y) for x,y in zip(a,b)), 'finite:', all(bool(torch.isfinite(x.float()).all()) for x in a))"
echo "== shapes A/B"
ONLY="@16" timeout 300 python tools/bench_shapes.py 2>&1 | grep "c3"
echo "-- UR_HIMG_WS=1"
ONLY="@16" UR_HIMG_WS=1 timeout 300 python tools/bench_shapes.py 2>&1 | grep "c3"
trace() {
  rm -rf $O/st
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/st -o st -- python tools/step_trace.py > $O/st.log 2>&1
  f=$(ls $O/st/*kernel_trace.csv $O/st/*/*kernel_trace.csv 2>/dev/null | head -1)
  python tools/step_trace.py --summarize $f > $O/r6_n_step_trace_$1.txt 2>&1
  head -1 $O/r6_n_step_trace_$1.txt; grep "halo_img" $O/r6_n_step_trace_$1.txt
  rm -rf $O/st
}
echo "== step trace default"; trace default
echo "== step trace UR_HIMG_WS=1"; UR_HIMG_WS=1 trace ws
echo "== forward A/B"
for i in 1 2; do
UR_HIMG_WS=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ws', d['ms_per_step'], d['output_finite'])"
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'])"
done
