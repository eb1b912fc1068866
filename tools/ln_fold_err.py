"""Op-level error of Linear(LayerNorm(x)) against fp64 of the STORED x: separate LayerNorm pass + GEMM vs the LayerNorm-folded GEMM."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn.functional as F
from unirestore_amd import ops
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
for dt in ("fp16", "bf16"):
    DT = ops.set_dtype(dt)
    for rows, c, n, k0 in [(4096, 320, 960, 320), (1024, 640, 640, 640), (256, 1280, 3840, 1280)]:
        g = torch.Generator().manual_seed(rows + n)
        x0 = torch.randn(rows, k0, generator=g); w0 = torch.randn(c, k0, generator=g) / math.sqrt(k0); r0 = torch.randn(rows, c, generator=g) * 0.5
        x = ops.linear(x0.to(DT).cuda(), ops.pack_conv(w0, None, "cuda"), residual=r0.to(DT).cuda(), rows=True)      # producer: leaves row sums
        xs = x.double().cpu()
        ga, be = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
        w = torch.randn(n, c, generator=g) / math.sqrt(c); b = 0.1 * torch.randn(n, generator=g)
        ref = F.linear(F.layer_norm(xs, (c,), ga.double(), be.double(), 1e-5), w.double(), b.double())
        y_sep = ops.linear(ops.layer_norm(x, ga.cuda(), be.cuda(), 1e-5), ops.pack_conv(w, b, "cuda"))
        y_fold = ops.linear(x, ops.pack_linear_ln(w, b, ga, be, 1e-5, "cuda"), ln_stats=ops.ln_of(x))
        # the same two with the weights' own rounding taken out of the reference (what is left is activation-side error)
        wr = ops.pack_conv(w, b, "cuda").w[:n, :c].double().cpu()
        ref_wr = F.linear(F.layer_norm(xs, (c,), ga.double(), be.double(), 1e-5), wr, b.double())
        wf = ops.pack_linear_ln(w, b, ga, be, 1e-5, "cuda").w[:n, :c].double().cpu()      # = round(W * gamma)
        ln0 = F.layer_norm(xs, (c,), None, None, 1e-5)
        ref_wf = F.linear(ln0, wf, (w.double() @ be.double()) + b.double())
        print(f"[{dt}] rows {rows} C {c} N {n}: separate {rel(y_sep, ref):.2e} (vs own rounded W {rel(y_sep, ref_wr):.2e})   folded {rel(y_fold, ref):.2e} (vs own rounded W*gamma {rel(y_fold, ref_wf):.2e})")
