"""In-graph timing of the transformer / SC-Tuner GEMM shapes (B=8, 512x512): python tools/ab_gemm.py   (UR_LIB=... for an A/B build)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from unirestore_amd import ops
from ab_micro import gtime  # noqa

shapes = [(32768, 320, 320, True, 560), (8192, 640, 640, True, 540), (2048, 1280, 1280, True, 540), (512, 1280, 1280, True, 160),
          (32768, 960, 320, False, 100), (8192, 1920, 640, False, 100), (2048, 3840, 1280, False, 100),
          (32768, 320, 1280, True, 100), (8192, 640, 2560, True, 100), (2048, 1280, 5120, True, 100),
          (32768, 320, 256, True, 60), (32768, 320, 640, False, 40), (8192, 640, 1920, False, 20), (2048, 1280, 2560, False, 40)]
tot = 0.0
for m, n, k, res, cnt in shapes:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(n, k) / k ** 0.5, torch.randn(n), "cuda")
    r = torch.randn(m, n, device="cuda").to(torch.bfloat16) if res else None
    us = gtime(lambda: ops.linear(x, pc, residual=r))
    byt = 2.0 * (m * k + n * k + m * n * (2 if res else 1))
    tot += us * cnt / 1e3
    print(f"M{m:6d} N{n:5d} K{k:5d} res={int(res)}  {us:7.1f} us  {2.0 * m * n * k / us / 1e6:6.1f} TF/s  {byt / us / 1e3:6.0f} GB/s   x{cnt} = {us * cnt / 1e3:6.2f} ms")
print(f"sum over listed launches: {tot:.1f} ms per forward")
