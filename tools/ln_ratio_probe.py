"""How ill-conditioned is the one-pass LayerNorm variance (E[x^2] - mean^2 in fp32) on the LayerNorm-folded GEMMs' inputs?
UR_CHAIN=0 python tools/ln_ratio_probe.py  -> per call: rows, C, median / max of mean^2 / var, and the error of the fp32 one-pass rstd."""
import os, sys
os.environ["UR_CHAIN"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, bench
from unirestore_amd import ops

dev = torch.device("cuda", 0)
m = bench.build_model(1, dev, 0, 1, dtype="fp16")
ref = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache", "fp16_attrib_ref.pt"))
orig = ops.linear
seen = []


def probe(x, pc, **kw):
    st = kw.get("ln_stats")
    if st is not None and len(seen) < 40:
        xf = x.reshape(-1, x.shape[-1]).double()
        mean, var = xf.mean(1), xf.var(1, unbiased=False)
        ratio = mean ** 2 / var
        plane, parts = st
        s = plane.view(parts, -1, 2).sum(0).double()
        c = xf.shape[1]
        mean1 = (s[:, 0] / c).float()
        var1 = (s[:, 1].float() / c - mean1 * mean1).clamp_min(0)              # the consumer's fp32 arithmetic
        rstd1 = torch.rsqrt(var1 + pc.ln_eps).double()
        rstd = torch.rsqrt(var + pc.ln_eps)
        err = ((rstd1 - rstd) / rstd).abs()
        seen.append(1)
        print(f"LN-folded GEMM rows {xf.shape[0]:6d} C {c:5d} -> N {pc.cout:5d}: mean^2/var median {float(ratio.median()):9.3g} max {float(ratio.max()):9.3g}   "
              f"rstd rel err median {float(err.median()):.2e} max {float(err.max()):.2e}   |x| max {float(xf.abs().max()):.1f} rms {float((xf ** 2).mean().sqrt()):.2f}")
    y = orig(x, pc, **kw)
    if st is not None and len(seen) <= 40:
        # kernel arithmetic against fp64 of the SAME operands (stored x, packed folded weights): only the output rounding should be left
        import torch.nn.functional as F
        xf = x.reshape(-1, x.shape[-1]).double()
        mean, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
        wd = pc.w.double()[:, :xf.shape[1]]
        z = (xf @ wd.t() - mean * wd.sum(1)[None]) * torch.rsqrt(var + pc.ln_eps) + pc.bias.double()[None]
        if pc.pair:
            zz = z.view(z.shape[0], -1, 2, 32)
            z = (zz[:, :, 0] * (F.gelu(zz[:, :, 1]) if kw.get("act") == ops.UR_ACT_GEGLU else zz[:, :, 1])).reshape(z.shape[0], -1)
        yy = y.reshape(-1, y.shape[-1]).double()
        ncmp = kw.get("n_split") or yy.shape[1]
        e = float((yy[:, :ncmp] - z[:, :ncmp]).norm() / z[:, :ncmp].norm())
        ev = ""
        if kw.get("yt") is not None:                       # transposed V columns: yt [B][C][T] vs z[:, n_split:]
            vt = kw["yt"].double()                          # [B, C, ldvt]
            b_, c_, _ = vt.shape
            t_ = kw["t_rows"]
            zv = z[:, ncmp:ncmp + c_].view(b_, t_, c_).permute(0, 2, 1)
            ev = f"  V^T rel-L2 {float((vt[:, :, :t_] - zv).norm() / zv.norm()):.2e}"
        print(f"    -> kernel vs fp64 of the same operands: rel-L2 {e:.2e}{ev}  (pair {pc.pair}, yt {kw.get('yt') is not None}, residual {kw.get('residual') is not None})")
    return y


ops.linear = probe
import unirestore_amd.modules.nn as nnm
nnm.ops.linear = probe
m._prepare()
m.base_model(ref["zt"], ref["ctl"], torch.tensor([999]))
