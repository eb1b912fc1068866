"""How fast does the vendor library run the square C->C GEMM shapes (plain x @ W^T, no epilogue)?   python tools/blas_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from ab_micro import gtime
for m, n, k in ((32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280), (512, 1280, 1280), (32768, 2560, 320), (2048, 1280, 5120)):
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    pc = ops.pack_conv(w.float().cpu(), None, "cuda")
    ub = gtime(lambda: torch.mm(x, w.t()))
    uo = gtime(lambda: ops.linear(x, pc))
    print(f"M{m} N{n} K{k}: vendor BLAS {ub:6.1f} us ({2.0 * m * n * k / ub / 1e6:6.0f} TF/s)   ours (no residual) {uo:6.1f} us")
