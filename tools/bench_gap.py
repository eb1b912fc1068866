"""Fixed per-launch cost inside a hipGraph: tiny problems, so the time is launch + dependent-latency chain of one tile."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.dirname(__file__))
import torch
from bench_one import gtime
from unirestore_amd import ops
x = torch.randn(1, 8, 8, 64, device="cuda").to(torch.bfloat16)
y = torch.empty(1, 8, 8, 64, device="cuda", dtype=torch.bfloat16)
r = torch.randn(1, 8, 8, 64, device="cuda").to(torch.bfloat16)
pcb = ops.pack_conv(torch.randn(64, 64, 1, 1) / 8, torch.randn(64), "cuda")
pcn = ops.pack_conv(torch.randn(64, 64, 1, 1) / 8, None, "cuda")
pc3 = ops.pack_conv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), "cuda")
a = torch.zeros(64, device="cuda")
print("torch add_ tiny           us/launch:", round(gtime(lambda: a.add_(1.0), reps=200), 2))
print("1x1 tiny, no bias         us/launch:", round(gtime(lambda: ops.conv(x, pcn, out=y), reps=200), 2))
print("1x1 tiny, bias            us/launch:", round(gtime(lambda: ops.conv(x, pcb, out=y), reps=200), 2))
print("1x1 tiny, bias+residual   us/launch:", round(gtime(lambda: ops.conv(x, pcb, out=y, residual=r), reps=200), 2))
print("3x3 tiny, bias            us/launch:", round(gtime(lambda: ops.conv(x, pc3, out=y), reps=200), 2))
g = torch.ones(64, device="cuda"); b = torch.zeros(64, device="cuda")
def gn():
    return ops.group_norm(x, g, b, 32, 1e-5, True)
print("GN tiny (stats+finalize+apply + arena fill) us:", round(gtime(gn, reps=100), 2))
q = torch.randn(1, 64, 192, device="cuda").to(torch.bfloat16); vt = torch.randn(1, 64, 64, device="cuda").to(torch.bfloat16)
print("attention tiny            us/launch:", round(gtime(lambda: ops.attention(q, q[:, :, 64:], vt, 1, 64, 64, 64, 0.125, ldq=192, ldk=192, bs_q=64*192, bs_k=64*192, bs_vt=64*64, batch=1), reps=200), 2))
for kk in (64, 320, 640, 1280, 2560):
    xk = torch.randn(1, 8, 8, kk, device="cuda").to(torch.bfloat16)
    pk = ops.pack_conv(torch.randn(64, kk, 1, 1) / kk ** 0.5, torch.randn(64), "cuda")
    print(f"1x1 tiny K={kk:5d} ({kk // 64:2d} k-tiles)  us/launch:", round(gtime(lambda: ops.conv(xk, pk, out=y), reps=200), 2))
for kk in (640, 1280):
    xk = torch.randn(2048, kk, device="cuda").to(torch.bfloat16)
    pk = ops.pack_conv(torch.randn(1280, kk, 1, 1) / kk ** 0.5, torch.randn(1280), "cuda")
    yk = torch.empty(2048, 1280, device="cuda", dtype=torch.bfloat16)
    print(f"M2048 N1280 K={kk} us/launch:", round(gtime(lambda: ops.linear(xk, pk), reps=50), 2))
