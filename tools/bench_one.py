import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unirestore_amd import ops


def gtime(f, reps=20):
    """GPU time per call: `reps` launches captured in one hipGraph (no host launch overhead), replayed 3x."""
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

B = 8
def run(name, h, w, cin, cout, k, res=False, act=0, pair=False, f32=False):
    x = torch.randn(B, h, w, cin, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.randn(cout), "cuda", pair=pair)
    r = torch.randn(B, h, w, pc.cout_out, device="cuda").to(torch.bfloat16) if res else None
    f = lambda: ops.conv(x, pc, residual=r, act=act, out_f32=f32)
    us = gtime(f)
    fl = 2.0 * B * h * w * cout * cin * k * k
    print(f"{name:40s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  out {B*h*w*pc.cout_out*2/1e6:.0f} MB")
if __name__ == "__main__": run("1x1 64->320 @64 (1 ktile)", 64, 64, 64, 320, 1)
if __name__ == "__main__": run("1x1 64->320 @64 f32out", 64, 64, 64, 320, 1, f32=True)
if __name__ == "__main__": run("1x1 320->320 @64 (5 ktiles)", 64, 64, 320, 320, 1)
if __name__ == "__main__": run("1x1 320->320 @64 +res", 64, 64, 320, 320, 1, res=True)
if __name__ == "__main__": run("1x1 640->320 @64 (10 ktiles)", 64, 64, 640, 320, 1)
if __name__ == "__main__": run("1x1 1280->320 @64 (20 ktiles)", 64, 64, 1280, 320, 1)
if __name__ == "__main__": run("1x1 2560->320 @64 (40 ktiles)", 64, 64, 2560, 320, 1)
if __name__ == "__main__": run("3x3 320->320 @64 (45 ktiles)", 64, 64, 320, 320, 3)
if __name__ == "__main__": run("1x1 320->2560 @64 none", 64, 64, 320, 2560, 1)
if __name__ == "__main__": run("1x1 320->2560 @64 geglu", 64, 64, 320, 2560, 1, act=ops.UR_ACT_GEGLU, pair=True)
if __name__ == "__main__": run("1x1 320->2560 @64 gate", 64, 64, 320, 2560, 1, act=ops.UR_ACT_GATE, pair=True)
if __name__ == "__main__": run("1x1 320->1280 @64 gelu", 64, 64, 320, 1280, 1, act=ops.UR_ACT_GELU)
if __name__ == "__main__": run("1x1 320->1280 @64 silu", 64, 64, 320, 1280, 1, act=ops.UR_ACT_SILU)
