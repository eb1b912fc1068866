import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unirestore_amd import ops
B = 8
x = torch.randn(B, 64, 64, 64, device="cuda").to(torch.bfloat16)
pc = ops.pack_conv(torch.randn(320, 64, 1, 1) / 8, torch.randn(320), "cuda")
f = lambda: ops.conv(x, pc)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("DBG", os.environ.get("UR_IGEMM_DBG"), "1x1 64->320@64:", round(e0.elapsed_time(e1) * 1e3 / 50, 1), "us")
