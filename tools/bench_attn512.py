"""d = 512 VAE mid-block attention: flash kernel vs the chunked GEMM form (in-graph).   python tools/bench_attn512.py"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from unirestore_amd.modules import nn as M
from ab_micro import gtime

for b, t in ((8, 4096), (1, 16384), (2, 2560)):
    c = 512
    qkv = torch.randn(b, t, 3 * c, device="cuda").to(torch.bfloat16)
    vt = torch.randn(b, c, t, device="cuda").to(torch.bfloat16)
    f = lambda: ops.attention(qkv, qkv[:, :, c:], vt, 1, c, t, t, 1 / math.sqrt(c), ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=b)
    g = lambda: M.attention_gemm(qkv[:, :, :c], qkv[:, :, c:2 * c], vt, 1, c, t)
    o1, o2 = f().float(), g().float()
    torch.cuda.synchronize()
    fl = 4.0 * b * t * t * c
    u1, u2 = gtime(f, reps=3), gtime(g, reps=3)
    print(f"B={b} T={t}: flash {u1:8.1f} us ({fl / u1 / 1e6:6.1f} TF/s)   chunked GEMM form {u2:8.1f} us   rel diff {float((o1 - o2).norm() / o2.norm()):.2e}")
