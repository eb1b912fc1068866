import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from ab_micro import gtime
for m, k, n, g in [(8, 512, 512, 1), (8, 2048, 2048, 16), (8, 2048, 16, 1), (8, 1280, 1280, 1), (20, 320, 1280, 1), (8, 256, 256, 1)]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k // g, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
    y = ops.linear_f32(x, w, b, groups=g)
    ref = torch.cat([x[:, i * (k // g):(i + 1) * (k // g)].double() @ w[i * (n // g):(i + 1) * (n // g)].double().t() for i in range(g)], 1) + b.double()
    us = gtime(lambda: ops.linear_f32(x, w, b, groups=g), reps=10)
    print(f"M{m} K{k} N{n} g{g}: {us:6.1f} us  err {float((y.double() - ref).abs().max()):.1e}")
