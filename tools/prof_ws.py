import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unirestore_amd import ops
B = 8
for (h, cin, cout, k) in [(64, 320, 320, 3), (64, 640, 640, 3), (64, 320, 2560, 1)]:
    x = torch.randn(B, h, h, cin, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.randn(cout), "cuda")
    for _ in range(3): ops.conv(x, pc)
    torch.cuda.synchronize()
    ws = ops.workspace(x.device)
    ws[:8].zero_()
    ops.conv(x, pc); torch.cuda.synchronize()
    v = ws[:8].cpu().tolist()
    nt = max(v[3], 1)
    print(f"{h}x{h} {cin}->{cout} k{k}: tiles {int(v[3])} | producer per tile: issue {v[0]/nt:.0f} wait {v[1]/nt:.0f} barrier {v[2]/nt:.0f} | consumer per tile: compute {v[4]/nt:.0f} barrier {v[5]/nt:.0f}  (clock64 ticks)")
