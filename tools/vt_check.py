import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from unirestore_amd import ops
for dt in ("bf16",):
    DT = ops.set_dtype(dt)
    for b, t, c in [(8, 64, 1280), (8, 256, 1280), (8, 64, 640), (2, 64, 1280), (8, 64, 320), (8, 128, 1280)]:
        g = torch.Generator().manual_seed(b * t + c)
        x0 = torch.randn(b * t, c, generator=g)
        w0 = torch.randn(c, c, generator=g) / math.sqrt(c)
        h2 = ops.linear(x0.to(DT).cuda(), ops.pack_conv(w0, None, "cuda"), rows=True)
        h = ops.carry(h2, h2.view(b, t, c))
        w = torch.randn(3 * c, c, generator=g) / math.sqrt(c)
        for use_ln in (False, True):
            vt = torch.empty((b, c, t), dtype=DT, device="cuda")
            if use_ln:
                pc = ops.pack_linear_ln(w, None, torch.ones(c), torch.zeros(c), 1e-5, "cuda")
                st = ops.ln_of(h)
                qk = ops.linear(h, pc, ln_stats=st, yt=vt, n_split=2 * c, t_rows=t)
                xin = torch.nn.functional.layer_norm(h.double().cpu(), (c,))
            else:
                pc = ops.pack_conv(w, None, "cuda")
                qk = ops.linear(h, pc, yt=vt, n_split=2 * c, t_rows=t)
                xin = h.double().cpu()
            z = xin.view(b * t, c) @ pc.w.double().cpu()[:, :c].t()
            zv = z[:, 2 * c:].view(b, t, c).permute(0, 2, 1)
            ev = float((vt.double().cpu() - zv).norm() / zv.norm())
            eq = float((qk.view(b * t, -1).double().cpu()[:, :2 * c] - z[:, :2 * c]).norm() / z[:, :2 * c].norm())
            print(f"[{dt}] B{b} T{t} C{c} ln={use_ln}: q|k rel-L2 {eq:.2e}  V^T rel-L2 {ev:.2e}")
