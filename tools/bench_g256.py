"""Large-N GEMM shapes of the UNet (GEGLU, fused QKV, ff2): time in a hipGraph, with/without the 256x256 tile kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from bench_one import gtime, B
from unirestore_amd import ops

def run(name, m, cin, cout, act=0, pair=False, res=False):
    x = torch.randn(m, cin, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, 1, 1) / cin ** 0.5, torch.randn(cout), "cuda", pair=pair)
    r = torch.randn(m, pc.cout_out, device="cuda").to(torch.bfloat16) if res else None
    us = gtime(lambda: ops.linear(x, pc, act=act, residual=r))
    print(f"{name:34s} {us:8.1f} us  {2.0*m*cin*cout/us/1e6:7.1f} TF/s")

G = ops.UR_ACT_GEGLU
run("geglu M32768 320->2560", 32768, 320, 2560, G, True)
run("geglu M8192  640->5120", 8192, 640, 5120, G, True)
run("geglu M2048 1280->10240", 2048, 1280, 10240, G, True)
run("qkv   M2048 1280->3840", 2048, 1280, 3840)
run("ff2   M32768 1280->320", 32768, 1280, 320, res=True)
run("lin   M32768 320->1280", 32768, 320, 1280)
run("lin   M8192 640->2560", 8192, 640, 2560)
run("lin   M32768 512->512", 32768, 512, 512)
