"""In-graph time of the small latency-chain kernels, alone and as the real dependent sequence:
producer conv (+ split-K reduce) -> gn_finalize -> gn_apply.   [UR_LIB=...] python tools/bench_fin.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from ab_micro import gtime

B = 8
for h, c, k in [(64, 320, 3), (32, 640, 3), (16, 1280, 3), (8, 1280, 3), (16, 1280, 1)]:
    x = torch.randn(B, h, h, c, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(c, c, k, k) / (c * k * k) ** 0.5, torch.randn(c), "cuda")
    ga, be = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    y = ops.conv(x, pc, gn=True)
    t_conv = gtime(lambda: ops.conv(x, pc, gn=True))
    t_fin = gtime(lambda: ops.gn_finalize(y, ga, be, 32, 1e-5))
    ab = ops.gn_finalize(y, ga, be, 32, 1e-5)
    t_app = gtime(lambda: ops.gn_apply(y, ab, silu=True))

    def seq():
        yy = ops.conv(x, pc, gn=True)
        return ops.gn_apply(yy, ops.gn_finalize(yy, ga, be, 32, 1e-5), silu=True)
    t_seq = gtime(seq)
    print(f"{h:3d}x{h:<3d} C{c:5d} k{k}: conv(+reduce) {t_conv:6.1f}  finalize {t_fin:5.1f}  apply {t_app:5.1f}  | sequence {t_seq:6.1f} us  (P = {ops.gn_of(y)[1]})")
