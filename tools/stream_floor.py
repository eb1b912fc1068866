"""What a pure streaming kernel needs for the traffic of the square C->C GEMMs (read x, read residual, write y):   python tools/stream_floor.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from ab_micro import gtime
for n, hw, c in ((8, 64, 320), (8, 32, 640), (8, 16, 1280)):
    x = torch.randn(n, hw, hw, c, device="cuda").to(torch.bfloat16)
    r = torch.randn(n, hw, hw, c, device="cuda").to(torch.bfloat16)
    s = torch.randn(n, c, device="cuda")
    us = gtime(lambda: ops.scale_channels(x, s, r))
    byt = 3 * x.numel() * 2
    pc = ops.pack_conv(torch.randn(c, c) / c ** 0.5, torch.randn(c), "cuda")
    xg, rg = x.view(-1, c), r.view(-1, c)
    ug = gtime(lambda: ops.linear(xg, pc, residual=rg))
    print(f"[{n}x{hw}x{hw}x{c}] x*s+r: {us:6.1f} us = {byt / us / 1e3:5.0f} GB/s    Linear {c}->{c} + residual: {ug:6.1f} us")
