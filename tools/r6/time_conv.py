"""us per launch of one 3x3 conv shape (graph of 10 launches): python tools/r6/time_conv.py cin cout hw [B]   (UR_LIB selects an A/B library)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from unirestore_amd import ops
cin, cout, hw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.randn(cout), "cuda")
f = lambda: ops.conv(x, pc)
for _ in range(2): f()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): f()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): g.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 30
print(f"{os.environ.get('UR_LIB', 'default'):40s} c3 {cin}->{cout}@{hw} B={B}: {us:9.1f} us  {2.0 * B * hw * hw * cout * cin * 9 / us / 1e6:7.1f} TF/s")
