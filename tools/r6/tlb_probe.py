"""Is the "cold" penalty of a per-layer GEMM (profiles/r6_k_cold_weights_prefetch.txt) cache misses or address-translation misses?
Between two launches of the SAME GEMM (same weights, same activations: everything cache-resident) a small kernel touches one 128-byte line
in each of P distinct 2-MiB pages (P x 128 B of cache footprint, P translations).  If the GEMM slows down with P it is the TLB.
python tools/r6/tlb_probe.py"""
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
    return e0.elapsed_time(e1) * 1e3 / reps


rows, c = 2048, 1280
pc = ops.pack_conv(torch.randn(c, c, 1, 1) / c ** 0.5, torch.randn(c), "cuda")
x = torch.randn(rows, c, device="cuda").to(torch.bfloat16)
r = torch.randn(rows, c, device="cuda").to(torch.bfloat16)
A = lambda: ops.linear(x, pc, residual=r)
big = torch.zeros(16 << 30, dtype=torch.uint8, device="cuda")              # 16 GiB: 8192 pages of 2 MiB
acc = torch.zeros(1, dtype=torch.float32, device="cuda")
N = 20
base = gtime_seq([A] * N) / N
print(f"GEMM 2048 x 1280 x 1280 alone: {base:6.2f} us")
for pages in (64, 512, 2048, 8192):
    v = big.view(-1, 2 << 20)[:pages, :4]                                  # 4 bytes in each of `pages` 2-MiB pages
    T = lambda v=v: acc.add_(v.float().sum())
    tt = gtime_seq([T] * N) / N
    both = gtime_seq([A, T] * N) / N
    print(f"toucher over {pages:5d} pages: alone {tt:6.2f} us;  (GEMM, toucher) pair {both:6.2f} us  ->  GEMM after the toucher {both - tt:6.2f} us (+{both - tt - base:5.2f})")
# the same count of lines inside few pages (cache footprint equal, translations few)
for lines in (8192,):
    v = big[:lines * 128].view(-1, 128)[:, :4]
    T = lambda v=v: acc.add_(v.float().sum())
    tt = gtime_seq([T] * N) / N
    both = gtime_seq([A, T] * N) / N
    print(f"toucher over {lines} lines of ONE MiB: alone {tt:6.2f} us;  pair {both:6.2f} us  ->  GEMM after the toucher {both - tt:6.2f} us (+{both - tt - base:5.2f})")
