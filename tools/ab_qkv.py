"""Fused QKV GEMM: cost of the transposed-V epilogue (in-graph).   python tools/ab_qkv.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from ab_micro import gtime
for b, t, c in ((8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (8, 64, 1280), (160, 4096, 256), (160, 1024, 256), (160, 256, 512)):
    x = torch.randn(b, t, c, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(3 * c, c) / c ** 0.5, torch.randn(3 * c), "cuda")
    vt = torch.empty(b, c, t, device="cuda", dtype=torch.bfloat16)
    u0 = gtime(lambda: ops.linear(x, pc))
    u1 = gtime(lambda: ops.linear(x, pc, yt=vt, n_split=2 * c, t_rows=t))
    print(f"B{b} T{t} C{c}: plain [B,T,3C] {u0:6.1f} us   with V^T epilogue {u1:6.1f} us")
