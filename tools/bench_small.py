"""Small/medium GEMM shapes of the 16x16 / 8x8 / 32x32 levels, timed in a hipGraph (rows=1: producer of LayerNorm row sums)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from bench_one import gtime
from unirestore_amd import ops

def run(m, cin, cout, rows=False):
    x = torch.randn(m, cin, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, 1, 1) / cin ** 0.5, torch.randn(cout), "cuda")
    r = torch.randn(m, cout, device="cuda").to(torch.bfloat16)
    def f():
        ops.linear(x, pc, residual=r, rows=rows)
    us = gtime(f)
    print(f"M{m} K{cin} N{cout} rows={int(rows)}  {us:7.1f} us  {2.0*m*cin*cout/us/1e6:7.1f} TF/s")

for rows in (False, True):
    run(2048, 1280, 1280, rows); run(512, 1280, 1280, rows); run(8192, 640, 640, rows); run(2048, 5120, 1280, rows); run(512, 5120, 1280, rows)
