#!/bin/bash
# Round 6, call 15: GroupNorm helper waves in the fused-GroupNorm halo conv (UR_HALO_NOGNH=1 = compute waves normalise, rounds 2-5)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== conv op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | grep -v amdgpu | tail -4
cat > /tmp/gn_case.py <<'PY'
import sys, math, torch; sys.path.insert(0, ".")
from unirestore_amd import ops
import torch.nn.functional as F
outs = []
for dt in ("bf16", "fp16"):
    ops.set_dtype(dt); DT = ops.act_dtype()
    for (n, cin, cout, h, w, ups) in [(2, 128, 128, 128, 128, False), (1, 256, 256, 64, 96, False), (2, 64, 128, 32, 64, True), (1, 192, 128, 48, 64, False)]:
        g = torch.Generator().manual_seed(cin + cout + h)
        x = (torch.randn(n, h, w, cin, generator=g) * 1.5 + 0.3).to(DT).cuda()
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(DT).float()
        pc = ops.pack_conv(wt, torch.randn(cout, generator=g), "cuda")
        ga, be = torch.randn(cin, generator=g).cuda(), torch.randn(cin, generator=g).cuda()
        assert ops.conv_plan(x, pc, upsample=ups, gn=True, gn_ab=True).prologue_ok
        ab = ops.gn_finalize(x, ga, be, 32, 1e-5)
        y = ops.conv(x, pc, upsample=ups, gn_ab=ab, gn_silu=True, gn=True)
        two = ops.conv(ops.gn_apply(x, ab, silu=True), pc, upsample=ups, gn=True)
        print(dt, (n, cin, cout, h, w, ups), "fused == two-pass:", torch.equal(y, two), flush=True)
        outs.append(y.cpu()); outs.append(ops.gn_of(y)[0].cpu())
torch.save(outs, sys.argv[1])
PY
echo "== fused GroupNorm: helpers"; timeout 300 python /tmp/gn_case.py /tmp/h.pt 2>&1 | grep -v amdgpu
echo "== fused GroupNorm: compute waves (UR_HALO_NOGNH=1)"; UR_HALO_NOGNH=1 timeout 300 python /tmp/gn_case.py /tmp/n.pt 2>&1 | grep -v amdgpu | tail -2
python -c "
import torch
a=torch.load('/tmp/h.pt'); b=torch.load('/tmp/n.pt')
print('tensors', len(a), 'helpers bit-identical to compute-wave normalisation:', all(torch.equal(x,y) for x,y in zip(a,b)))"
cat > /tmp/gn_time.py <<'PY'
import sys, math, torch; sys.path.insert(0, "."); sys.path.insert(0, "tools")
from unirestore_amd import ops
from bench_one import gtime
for (n, cin, cout, hw) in [(8, 128, 128, 512), (8, 256, 256, 256), (8, 128, 256, 256)]:
    x = torch.randn(n, hw, hw, cin, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.randn(cout), "cuda")
    ab = ops.gn_finalize(x, torch.randn(cin, device="cuda"), torch.randn(cin, device="cuda"), 32, 1e-5)
    t_f = gtime(lambda: ops.conv(x, pc, gn_ab=ab, gn_silu=True, gn=True))
    t_p = gtime(lambda: ops.conv(x, pc, gn=True))
    t_a = gtime(lambda: ops.gn_apply(x, ab, silu=True))
    print(f"c3 {cin}->{cout}@{hw} B={n}: fused GroupNorm {t_f:8.1f} us   plain conv {t_p:8.1f} us   separate apply pass {t_a:7.1f} us")
PY
echo "== timing: helpers"; timeout 300 python /tmp/gn_time.py 2>&1 | grep c3
echo "== timing: UR_HALO_NOGNH=1"; UR_HALO_NOGNH=1 timeout 300 python /tmp/gn_time.py 2>&1 | grep c3
echo "== forward A/B"
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('helpers', d['ms_per_step'], d['output_finite'])"
UR_HALO_NOGNH=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-fp16 --no-profile --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no helpers', d['ms_per_step'])"
done
