import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.dirname(__file__))
import torch
from unirestore_amd import ops
from bench_one import gtime
B = 8
for (h, c) in [(64, 320), (64, 640), (64, 960), (32, 640), (32, 1280), (32, 1920), (16, 1280), (8, 1280), (512, 128), (256, 256)]:
    x0 = torch.randn(B, h, h, c, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.eye(c).view(c, c, 1, 1), None, "cuda")
    x = ops.conv(x0, pc, gn=True)           # leaves fused / fallback channel sums on x
    ga, be = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    us = gtime(lambda: ops.group_norm(x, ga, be, 32, 1e-5, True))
    mb = x.numel() * 2 / 1e6
    print(f"GN(apply-only) {h}x{h}x{c}: {us:8.1f} us  {mb:7.1f} MB  {2 * mb / us / 1e3:.2f} TB/s")
