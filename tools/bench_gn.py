import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from unirestore_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
from bench_one import gtime
B = 8
for (h, c) in [(64, 320), (64, 640), (64, 960), (32, 640), (32, 1280), (32, 1920), (16, 1280), (16, 2560), (8, 1280), (8, 2560), (512, 128), (256, 256)]:
    x = torch.randn(B, h, h, c, device="cuda").to(torch.bfloat16)
    ga, be = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    us = gtime(lambda: ops.group_norm(x, ga, be, 32, 1e-5, True))
    mb = x.numel() * 2 / 1e6
    print(f"GN {h}x{h}x{c}: {us:8.1f} us  {mb:7.1f} MB  eff {3 * mb / us / 1e3:.2f} TB/s")
for (rows, c) in [(32768, 320), (8192, 640), (2048, 1280)]:
    x = torch.randn(rows, c, device="cuda").to(torch.bfloat16)
    ga, be = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    us = gtime(lambda: ops.layer_norm(x, ga, be, 1e-5))
    mb = x.numel() * 2 / 1e6
    print(f"LN {rows}x{c}: {us:8.1f} us  {mb:7.1f} MB  eff {2 * mb / us / 1e3:.2f} TB/s")
x = torch.randn(2, 4, 4, 2560).to(torch.bfloat16).float()
ga, be = torch.randn(2560), torch.randn(2560)
ref = F.silu(F.group_norm(x.permute(0, 3, 1, 2), 32, ga, be, eps=1e-5)).permute(0, 2, 3, 1)
y = ops.group_norm(x.to(torch.bfloat16).cuda(), ga.cuda(), be.cuda(), 32, 1e-5, True).float().cpu()
print("gn2560 err", float((y - ref).norm() / ref.norm()), float((y - ref).abs().max()))
