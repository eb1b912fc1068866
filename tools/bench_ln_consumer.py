"""LayerNorm-folded consumer GEMM vs the same GEMM without the fold, in a hipGraph (cost of gathering the row-sum planes)."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.dirname(__file__))
import torch
from bench_one import gtime
from unirestore_amd import ops

def run(m, c, n, pair=False):
    x0 = torch.randn(m, c, device="cuda").to(torch.bfloat16)
    pc0 = ops.pack_conv(torch.randn(c, c, 1, 1) / math.sqrt(c), torch.randn(c), "cuda")
    x = ops.linear(x0, pc0, rows=True)
    st, parts = ops.ln_of(x)
    w, b = torch.randn(n, c) / math.sqrt(c), torch.randn(n)
    pln = ops.pack_linear_ln(w, b, torch.ones(c), torch.zeros(c), 1e-5, "cuda", pair=pair)
    ppl = ops.pack_conv(w.view(n, c, 1, 1), b, "cuda", pair=pair)
    act = ops.UR_ACT_GEGLU if pair else ops.UR_ACT_NONE
    t_ln = gtime(lambda: ops.linear(x, pln, ln_stats=(st, parts), act=act))
    t_pl = gtime(lambda: ops.linear(x, ppl, act=act))
    print(f"M{m} C{c} N{n} pair={int(pair)} parts={parts}:  LN-folded {t_ln:6.1f} us   plain {t_pl:6.1f} us")

run(2048, 1280, 1280); run(2048, 1280, 3840); run(2048, 1280, 10240, True)
run(8192, 640, 640); run(8192, 640, 1920); run(8192, 640, 5120, True)
run(32768, 320, 320); run(32768, 320, 960); run(32768, 320, 2560, True)
print("-- N sweep at M32768 C320")
for n in (480, 640, 800, 960, 1120, 1280, 1600):
    run(32768, 320, n)
