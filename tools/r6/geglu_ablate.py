"""GEGLU GEMM (gemm_glds_kernel<256,320,8,1,2,pair>) with the timing-only ablation bits of UR_IGEMM_DBG (results wrong by design):
   for d in 0 1 2 4 8; do UR_IGEMM_DBG=$d python tools/r6/geglu_ablate.py; done      (1 no DMA, 2 no MFMA, 4 no epilogue, 8 no K loop)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from unirestore_amd import ops


def gtime(f, reps=20):
    for _ in range(2): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


for rows, cin, hid in ((8192, 640, 2560), (2048, 1280, 5120), (32768, 320, 1280)):
    pc = ops.pack_conv(torch.randn(2 * hid, cin) / cin ** 0.5, torch.randn(2 * hid), "cuda", pair=True)
    x = torch.randn(rows, cin, device="cuda").to(torch.bfloat16)
    t = gtime(lambda: ops.linear(x, pc, act=ops.UR_ACT_GEGLU))
    print(f"DBG={os.environ.get('UR_IGEMM_DBG', '0')}  GEGLU {rows} x {cin} -> 2 x {hid}: {t:7.1f} us  {2.0 * rows * cin * 2 * hid / t / 1e6:7.1f} TF/s")
