"""Does a small pure GEMM give the same bits when another kernel runs beside it?  python tools/race_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from unirestore_amd import ops
torch.manual_seed(0)
dev = "cuda"
B = 8
def mk(hw, cin, cout, k, res):
    x = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.randn(cout), dev)
    r = torch.randn(B, hw, hw, cout, device=dev).to(torch.bfloat16) if res else None
    return x, pc, r
side_cases = {"g 256->1280 @8 res": mk(8, 256, 1280, 1, True), "g 1280->1280 @8 gelu": mk(8, 1280, 1280, 1, False),
              "g 640->640 @32 res": mk(32, 640, 640, 1, True), "g 256->320 @64 res": mk(64, 256, 320, 1, True), "g 320->320 @64": mk(64, 320, 320, 1, False)}
main_cases = [mk(8, 1280, 1280, 3, True), mk(16, 1280, 1280, 3, True), mk(64, 320, 320, 3, False), mk(32, 640, 640, 1, True)]
side = torch.cuda.Stream()
for name, (x, pc, r) in side_cases.items():
    ref = ops.conv(x, pc, residual=r).clone()
    torch.cuda.synchronize()
    bad = 0
    for it in range(40):
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            outs = [ops.conv(x, pc, residual=r) for _ in range(6)]
        for mx, mpc, mr in main_cases:
            ops.conv(mx, mpc, residual=mr, gn=True)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    alone = sum(0 if torch.equal(ops.conv(x, pc, residual=r), ref) else 1 for _ in range(50))
    print(f"{name:26s} mismatches beside other kernels: {bad}/240   alone: {alone}/50")

# ---- the other direction: main-stream kernels of the mid / up path while a side stream spams small GEMMs -----------------------
print("main-stream ops beside side-stream GEMMs:")
sx, spc, sr = side_cases["g 640->640 @32 res"]
sx2, spc2, sr2 = side_cases["g 256->320 @64 res"]
def spam():
    for _ in range(4):
        ops.conv(sx, spc, residual=sr); ops.conv(sx2, spc2, residual=sr2, gn=True)
main_ops = {}
x8, pc8, r8 = mk(8, 1280, 1280, 3, True); main_ops["c3 1280@8 split+gn-reduce"] = lambda: ops.conv(x8, pc8, residual=r8, gn=True)
x16, pc16, r16 = mk(16, 1280, 1280, 3, True); main_ops["c3 1280@16 himg split"] = lambda: ops.conv(x16, pc16, residual=r16, gn=True)
x32, pc32, r32 = mk(32, 640, 640, 3, True); main_ops["c3 640@32 halo split"] = lambda: ops.conv(x32, pc32, residual=r32, gn=True)
xg = torch.randn(B * 256, 1280, device=dev).to(torch.bfloat16)
pcg = ops.pack_conv(torch.randn(10240, 1280) / 36, torch.randn(10240), dev, pair=True); main_ops["geglu 1280->10240 M2048"] = lambda: ops.linear(xg, pcg, act=ops.UR_ACT_GEGLU)
pcq = ops.pack_conv(torch.randn(1280, 1280) / 36, None, dev); main_ops["linear 1280 M2048 rows"] = lambda: ops.linear(xg, pcq, rows=True)
t, heads, d = 256, 20, 64
c = heads * d
qkv = torch.randn(B, t, 3 * c, device=dev).to(torch.bfloat16); vt = torch.randn(B, c, t, device=dev).to(torch.bfloat16)
main_ops["attention T256 h20"] = lambda: ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, 0.125, ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=B)
ga, be = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
main_ops["group_norm 1280@16"] = lambda: ops.group_norm(x16, ga, be, 32, 1e-5, True)
for name, f in main_ops.items():
    ref = f().clone(); torch.cuda.synchronize()
    bad = 0
    for it in range(40):
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            spam()
        outs = [f() for _ in range(4)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    print(f"{name:28s} mismatches beside side-stream GEMMs: {bad}/160")
