"""In-graph micro-benchmark of a few hot shapes (works on the r1 and r2 trees: python tools/ab_micro.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import math, torch
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
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


if __name__ == "__main__":
    B = 8
    def conv(name, hw, cin, cout, k, gn=False, res=False, n=B):
        x = torch.randn(n, hw, hw, cin, device="cuda").to(torch.bfloat16)
        pc = ops.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.randn(cout), "cuda")
        r = torch.randn(n, hw, hw, cout, device="cuda").to(torch.bfloat16) if res else None
        def f():
            if hasattr(ops, "arena"): ops.arena().reset()
            return ops.conv(x, pc, residual=r, gn=gn)
        us = gtime(f)
        print(f"{name:44s} {us:8.1f} us  {2.0 * n * hw * hw * cout * cin * k * k / us / 1e6:7.1f} TF/s")

    conv("c3 320->320 @64 plain", 64, 320, 320, 3)
    conv("c3 320->320 @64 gn+res", 64, 320, 320, 3, gn=True, res=True)
    conv("c3 640->640 @32 gn+res", 32, 640, 640, 3, gn=True, res=True)
    conv("c3 1280->1280 @16 gn+res", 16, 1280, 1280, 3, gn=True, res=True)
    conv("c3 1280->1280 @8 gn+res", 8, 1280, 1280, 3, gn=True, res=True)
    conv("c3 128->128 @512 gn (VAE)", 512, 128, 128, 3, gn=True, n=2)
    conv("g1 320->320 @64 res", 64, 320, 320, 1, res=True)
    conv("g1 640->640 @32 res", 32, 640, 640, 1, res=True)
    conv("g1 1280->1280 @16 res", 16, 1280, 1280, 1, res=True)
    conv("g1 320->320 @64 gn+res", 64, 320, 320, 1, gn=True, res=True)
    t, heads, d = 4096, 5, 64
    c = heads * d
    qkv = torch.randn(B, t, 3 * c, device="cuda").to(torch.bfloat16)
    vt = torch.randn(B, c, t, device="cuda").to(torch.bfloat16)
    us = gtime(lambda: ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, 0.125, ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=B), reps=5)
    print(f"{'attention T=4096 h5 d64':44s} {us:8.1f} us  {4.0 * B * heads * t * t * d / us / 1e6:7.1f} TF/s")

    if hasattr(ops, "gn_finalize"):          # r2: GroupNorm apply fused into the conv's loader vs a separate apply pass
        for hw, cch, n in ((64, 320, B), (32, 640, B), (16, 1280, B), (512, 128, 2)):
            x = torch.randn(n, hw, hw, cch, device="cuda").to(torch.bfloat16)
            pc = ops.pack_conv(torch.randn(cch, cch, 3, 3) / (cch * 9) ** 0.5, torch.randn(cch), "cuda")
            ga, be = torch.ones(cch, device="cuda"), torch.zeros(cch, device="cuda")
            plane = ops.gn_partials(x)
            x._gn = plane
            ab = ops.gn_finalize(x, ga, be, 32, 1e-5)
            t_fin = gtime(lambda: ops.gn_finalize(x, ga, be, 32, 1e-5))
            t_app = gtime(lambda: ops.gn_apply(x, ab, silu=True))
            xn = ops.gn_apply(x, ab, silu=True)
            t_conv = gtime(lambda: ops.conv(xn, pc, gn=True))
            t_fused = gtime(lambda: ops.conv(x, pc, gn_ab=ab, gn_silu=True, gn=True))
            print(f"GN+conv {cch}@{hw} n={n}: finalize {t_fin:.1f}  apply {t_app:.1f}  conv {t_conv:.1f}  fused conv {t_fused:.1f} us   (apply+conv {t_app + t_conv:.1f})")
