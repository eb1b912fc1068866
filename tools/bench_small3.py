"""3x3 convs of the 16x16 / 8x8 levels (split-K regime), timed in a hipGraph."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from bench_one import gtime
from unirestore_amd import ops

def run(b, hw, cin, cout, gn=True):
    x = torch.randn(b, hw, hw, cin, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout), "cuda")
    def f():
        ops.conv(x, pc, gn=gn)
    us = gtime(f)
    print(f"c3 B{b} {hw}x{hw} {cin}->{cout}  {us:7.1f} us  {2.0*b*hw*hw*cin*cout*9/us/1e6:7.1f} TF/s")

run(8, 16, 1280, 1280); run(8, 8, 1280, 1280); run(8, 16, 2560, 1280); run(8, 8, 2560, 1280); run(8, 16, 640, 1280); run(8, 32, 640, 640); run(8, 32, 1280, 640)
