import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, torch.nn.functional as F
from unirestore_amd import ops
from golden_util import rel_l2
g = torch.Generator().manual_seed(0)
for (n, cin, cout, h, w, k) in [(1, 320, 320, 32, 32, 3), (2, 64, 128, 16, 16, 3), (2, 1280, 1280, 8, 8, 3), (1, 128, 256, 24, 8, 1)]:
    x = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).float(); wt = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(torch.bfloat16).float()
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, padding=k // 2)
    pc = ops.pack_conv(wt, b, "cuda")
    y = ops.conv(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda(), pc)
    yy = y.float().cpu().permute(0, 3, 1, 2)
    print((n, cin, cout, h, w, k), "kcm", pc.kcm, "err", rel_l2(yy, ref), "nan", bool(torch.isnan(yy).any()))
