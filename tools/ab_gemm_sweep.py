"""Per-shape sweep of the g1 tile shape x ring depth variants (A/B library from tools/ab_variants.sh):
   UR_LIB=unirestore_amd/build_ab/libur_ab.so python tools/ab_gemm_sweep.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops
from ab_micro import gtime

NAMES = ["64x64/2", "64x64/3", "64x64/4", "128x64/2", "128x64/3", "128x128/2", "128x128/3", "128x160/2", "128x160/3", "128x320/2",
         "64x128/2", "64x128/4", "64x320/2", "64x160/2", "64x160/4", "64x64/6", "256x128/2", "128x128/4"]
shapes = [(32768, 320, 320, True, 560), (8192, 640, 640, True, 540), (2048, 1280, 1280, True, 540), (512, 1280, 1280, True, 160),
          (32768, 960, 320, False, 100), (8192, 1920, 640, False, 100), (2048, 3840, 1280, False, 100),
          (32768, 320, 1280, True, 100), (8192, 640, 2560, True, 100), (2048, 1280, 5120, True, 100),
          (32768, 320, 256, True, 60), (32768, 320, 640, False, 40), (8192, 640, 1920, False, 20), (2048, 1280, 2560, False, 40),
          (512, 1280, 2560, False, 60), (512, 1280, 256, True, 60)]
tot0 = tot1 = 0.0
for m, n, k, res, cnt in shapes:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(n, k) / k ** 0.5, torch.randn(n), "cuda")
    r = torch.randn(m, n, device="cuda").to(torch.bfloat16) if res else None
    os.environ["UR_AB_ID"] = "-1"
    ref = ops.linear(x, pc, residual=r).float()
    gtime(lambda: ops.linear(x, pc, residual=r))
    base = gtime(lambda: ops.linear(x, pc, residual=r))
    row = []
    for i, nm in enumerate(NAMES):
        if n % 160 and "160" in nm.split("/")[0] or n % 320 and "320" in nm.split("/")[0]:
            continue
        os.environ["UR_AB_ID"] = str(i)
        try:
            y = ops.linear(x, pc, residual=r).float()
        except Exception as e:      # noqa
            continue
        err = float((y - ref).abs().max())
        us = gtime(lambda: ops.linear(x, pc, residual=r))
        row.append((us, nm, err))
    os.environ["UR_AB_ID"] = "-1"
    base2 = gtime(lambda: ops.linear(x, pc, residual=r))
    row.sort()
    best = row[0]
    base = min(base, base2)
    tot0 += base * cnt / 1e3
    tot1 += min(best[0], base) * cnt / 1e3
    print(f"M{m:6d} N{n:5d} K{k:5d} res={int(res)} default {base:6.1f}/{base2:6.1f} us | " + "  ".join(f"{nm} {us:.1f}{'' if err < 1e-2 else '!ERR'}" for us, nm, err in row[:6]))
print(f"per forward: default {tot0:.1f} ms, best-of {tot1:.1f} ms")
