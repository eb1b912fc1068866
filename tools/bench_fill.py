"""GEMM shapes whose 256-row tiling under-fills 256 CUs, in a hipGraph (UR_IGEMM_NOFILL=1 for the old dispatch)."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.dirname(__file__))
import torch
from bench_one import gtime
from unirestore_amd import ops
for (m, k, n, res) in [(8192, 2560, 640, True), (8192, 1920, 640, False), (8192, 1280, 640, False), (8192, 960, 640, False),
                       (2048, 1280, 3840, False), (2048, 2560, 1280, False), (2048, 1920, 1280, False)]:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(n, k, 1, 1) / math.sqrt(k), torch.randn(n), "cuda")
    r = torch.randn(m, n, device="cuda").to(torch.bfloat16) if res else None
    us = gtime(lambda: ops.linear(x, pc, residual=r))
    print(f"M{m} K{k} N{n}: {us:7.1f} us  {2.0*m*k*n/us/1e6:7.1f} TF/s")
