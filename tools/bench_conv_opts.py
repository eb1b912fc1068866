import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.dirname(__file__))
import torch
from unirestore_amd import ops
from bench_one import gtime
B = 8
for (h, cin, cout) in [(64, 320, 320), (32, 640, 640), (16, 1280, 1280)]:
    x = torch.randn(B, h, h, cin, device="cuda").to(torch.bfloat16)
    r = torch.randn(B, h, h, cout, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.randn(cout), "cuda")
    bias_row = torch.randn(cout, device="cuda")
    def run(**kw):
        def f():
            return ops.conv(x, pc, **kw)
        return gtime(f)
    print(f"{h}x{h} {cin}->{cout}: plain {run():.1f}  +res {run(residual=r):.1f}  +gn {run(gn=True):.1f}  +res+gn {run(residual=r, gn=True):.1f}  +bias-row {run(bias=bias_row):.1f} us")
