"""How much of a per-layer launch is the COLD weight fetch?  The same launch replayed in a graph with (a) one weight tensor (warm: it stays in
L2 / the 256-MB Infinity Cache) and (b) a ring of distinct weight tensors larger than the Infinity Cache (cold: every launch streams its weights
from HBM, as in the real forward where a UNet step touches 1.7 GB of weights).   python tools/cold_weights.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from unirestore_amd import ops


def gtime_seq(fs, reps=3):
    for f in fs[:2]: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fs: f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fs))


B = 8
cases = [("linear 1280->1280 @ 2048 rows", None, 2048, 1280, 1280, 1), ("linear 640->640 @ 8192 rows", None, 8192, 640, 640, 1),
         ("linear 1280->1280 @ 512 rows", None, 512, 1280, 1280, 1), ("conv3x3 320->320 @ 64x64", 64, 0, 320, 320, 3),
         ("conv3x3 640->640 @ 32x32", 32, 0, 640, 640, 3), ("conv3x3 1280->1280 @ 16x16", 16, 0, 1280, 1280, 3), ("conv3x3 1280->1280 @ 8x8", 8, 0, 1280, 1280, 3)]
for name, hw, rows, cin, cout, k in cases:
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    wbytes = cout * cin * k * k * 2
    nsets = max(4, int(600e6 // wbytes) + 1)                      # > 2x the Infinity Cache
    pcs = [ops.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.randn(cout), "cuda") for _ in range(nsets)]
    if hw:
        x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
        warm = [lambda: ops.conv(x, pcs[0], gn=True)] * nsets
        cold = [(lambda p: (lambda: ops.conv(x, p, gn=True)))(p) for p in pcs]
    else:
        x = torch.randn(rows, cin, device="cuda").to(torch.bfloat16)
        r = torch.randn(rows, cout, device="cuda").to(torch.bfloat16)
        warm = [lambda: ops.linear(x, pcs[0], residual=r)] * nsets
        cold = [(lambda p: (lambda: ops.linear(x, p, residual=r)))(p) for p in pcs]
    # (c) "mall": a ring of ~100 MB - larger than the eight 4-MB L2s together, well inside the 256-MB Infinity Cache: what a prefetch
    #     of the next layer's weights into the Infinity Cache could give back of the cold penalty
    nmall = max(2, min(nsets, int(100e6 // wbytes)))
    mall = (cold[:nmall] * (nsets // nmall + 1))[:nsets]
    tw, tm, tc = gtime_seq(warm), gtime_seq(mall), gtime_seq(cold)
    print(f"{name:34s} weights {wbytes / 1e6:6.1f} MB x {nsets:4d} sets: warm {tw:7.1f} us   infinity-cache ring ({nmall} sets) {tm:7.1f} us   cold {tc:7.1f} us   "
          f"(+{tc - tw:5.1f} us = {wbytes / max(tc - tw, 1e-3) / 1e6:6.2f} TB/s for the weights alone)")
    del pcs
    torch.cuda.empty_cache()
