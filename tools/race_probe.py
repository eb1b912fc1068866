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
