"""Why does a per-layer GEMM take ~5 us longer inside the step graph than in a back-to-back loop of itself?  Not the weights
(profiles/r6_k_cold_weights_prefetch.txt).  This probe times graphs of 40 launches: one GEMM shape alone (A A A ..), another alone (B B B ..),
the two alternating (A B A B ..: every launch follows a DIFFERENT kernel: cold instruction cache), and A with distinct activation tensors
per launch (a ring inside the Infinity Cache / beyond it).   python tools/r6/kernel_switch.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from unirestore_amd import ops


def gtime_seq(fs, reps=3):
    for f in fs[:4]: f()
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


def lin(rows, cin, cout, nx=1):
    pc = ops.pack_conv(torch.randn(cout, cin, 1, 1) / cin ** 0.5, torch.randn(cout), "cuda")
    xs = [torch.randn(rows, cin, device="cuda").to(torch.bfloat16) for _ in range(nx)]
    rs = [torch.randn(rows, cout, device="cuda").to(torch.bfloat16) for _ in range(nx)]
    return [(lambda x, r: (lambda: ops.linear(x, pc, residual=r)))(x, r) for x, r in zip(xs, rs)]


N = 40
A = lin(2048, 1280, 1280)[0]            # gemm_glds_kernel<64,64,2,2,2>
B = lin(8192, 640, 640)[0]              # gemm_glds_kernel<128,64,2,2,2>
C = lin(32768, 320, 320)[0]             # another tile shape again
x4 = torch.randn(8, 16, 16, 1280, device="cuda").to(torch.bfloat16)
ga, be = torch.randn(1280, device="cuda"), torch.randn(1280, device="cuda")
G = lambda: ops.group_norm(x4, ga, be, 32, 1e-5, True)       # stats + finalize + apply: three small kernels
ta, tb, tc = gtime_seq([A] * N), gtime_seq([B] * N), gtime_seq([C] * N)
tab = gtime_seq([A, B] * (N // 2))
tabc = gtime_seq([A, B, C] * 13)
tg = gtime_seq([G] * N)
tag = gtime_seq([A, G] * (N // 2))
print(f"A alone {ta:6.2f} us   B alone {tb:6.2f} us   C alone {tc:6.2f} us")
print(f"A B A B ..   {tab:6.2f} us per launch  vs mean(A, B) {(ta + tb) / 2:6.2f}  -> +{tab - (ta + tb) / 2:5.2f} us per kernel switch")
print(f"A B C A B C  {tabc:6.2f} us per launch  vs mean       {(ta + tb + tc) / 3:6.2f}  -> +{tabc - (ta + tb + tc) / 3:5.2f} us")
print(f"G alone (GroupNorm = 3 small kernels) {tg:6.2f} us;  A G A G .. {tag:6.2f} per item vs mean {(ta + tg) / 2:6.2f} -> +{2 * (tag - (ta + tg) / 2):5.2f} us per (A, G) pair")
for nx, what in ((16, "16 activation sets (170 MB with residuals/outputs: inside the Infinity Cache)"), (64, "64 sets (1 GB: beyond it)")):
    fs = lin(2048, 1280, 1280, nx)
    t = gtime_seq((fs * (N // nx + 1))[:max(N, nx)])
    print(f"A over {what}: {t:6.2f} us per launch")

# ---- producer -> consumer: does a GEMM read what the PREVIOUS launch wrote as fast as a tensor nobody wrote? ----------------------------
rows, c = 2048, 1280
pcs = [ops.pack_conv(0.7 * torch.randn(c, c, 1, 1) / c ** 0.5, 0.1 * torch.randn(c), "cuda") for _ in range(4)]      # contractive: values stay O(1)
r0 = torch.randn(rows, c, device="cuda").to(torch.bfloat16)
for nb, what in ((2, "2 ping-pong buffers"), (8, "8 buffers")):
    bufs = [torch.randn(rows, c, device="cuda").to(torch.bfloat16) for _ in range(nb)]
    fs = [(lambda i: (lambda: ops.linear(bufs[(i - 1) % nb], pcs[i % 4], residual=r0, out=bufs[i % nb].view(1, 1, rows, c))))(i) for i in range(N)]
    t = gtime_seq(fs)
    ok = bool(torch.isfinite(bufs[0].float()).all()) and float(bufs[0].float().abs().mean()) > 1e-3
    print(f"chain y[i] = Linear_i(y[i-1]) + r0, {what}: {t:6.2f} us per launch (values finite and non-zero: {ok})   [A alone, same shape: {ta:6.2f}]")
# same chain, but the residual is what the launch before the previous one wrote (the transformer's pattern)
bufs = [torch.randn(rows, c, device="cuda").to(torch.bfloat16) * 0.1 for _ in range(3)]
pch = [ops.pack_conv(0.4 * torch.randn(c, c, 1, 1) / c ** 0.5, None, "cuda") for _ in range(4)]
fs = [(lambda i: (lambda: ops.linear(bufs[(i - 1) % 3], pch[i % 4], residual=bufs[(i - 2) % 3], out=bufs[i % 3].view(1, 1, rows, c))))(i) for i in range(12)]
t = gtime_seq(fs)
print(f"chain y[i] = Linear_i(y[i-1]) + y[i-2], 3 buffers, 12 launches: {t:6.2f} us per launch (finite: {bool(torch.isfinite(bufs[0].float()).all())})")
