"""dwconv3x3 (+SimpleGate): strip kernel vs one-pixel kernel.  UR_DW_NOSTRIP=1 python tools/ab_dw.py ; python tools/ab_dw.py  (prints in-graph us + a checksum)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from ab_micro import gtime
for n, h, c, cnt in [(8, 256, 256, 2), (8, 128, 512, 2), (8, 64, 1024, 10)]:
    g = torch.Generator().manual_seed(h + c)
    x = torch.randn(n, h, h, c, generator=g).to(torch.bfloat16).cuda()
    w = torch.randn(9, c, generator=g).cuda(); b = torch.randn(c, generator=g).cuda()
    y = ops.dwconv3x3(x, w, b, gate=True)
    us = gtime(lambda: ops.dwconv3x3(x, w, b, gate=True), reps=5)
    print(f"N{n} {h}x{h} C{c} gate: {us:7.1f} us  x{cnt}   checksum {int(y.view(torch.int16).to(torch.int64).sum())}")
