import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.dirname(__file__))
import torch
from unirestore_amd import ops
from bench_one import gtime
B = 8
for (t, heads, d, tk) in [(4096, 5, 64, 4096), (4096, 4, 64, 4096), (1024, 10, 64, 1024), (4096, 5, 64, 77), (256, 20, 64, 256), (256, 4, 128, 256)]:
    c = heads * d
    q = torch.randn(B, t, 3 * c, device="cuda").to(torch.bfloat16)
    ldvt = (tk + 7) // 8 * 8
    if tk == t:
        vt = torch.randn(B, c, ldvt, device="cuda").to(torch.bfloat16)
        f = lambda: ops.attention(q, q[:, :, c:], vt, heads, d, t, tk, 1 / math.sqrt(d), ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * ldvt, batch=B)
    else:
        k = torch.randn(1, tk, 2 * c, device="cuda").to(torch.bfloat16); vt = torch.zeros(1, c, ldvt, device="cuda", dtype=torch.bfloat16); vt[:, :, :tk] = torch.randn(1, c, tk)
        f = lambda: ops.attention(q, k, vt, heads, d, t, tk, 1 / math.sqrt(d), ldq=3 * c, ldk=2 * c, bs_q=t * 3 * c, bs_k=0, bs_vt=0, batch=B)
    us = gtime(f)
    print(f"attn T={t} Tk={tk} H={heads} D={d}: {us:8.1f} us  {4.0 * B * heads * t * tk * d / us / 1e6:7.1f} TF/s")
