"""Token-stationary fused chains (csrc/tchain.hip) vs plain PyTorch fp32 on the same 16-bit-rounded inputs, both 16-bit types,
and vs the per-layer HIP path they replace (same weights).  Semantics: diffusers BasicTransformerBlock as reached through
/root/reference/src/modules/diffuie/base_model.py:137-160,184-198.

Tolerances (stated): LayerNorm-folded GEMM chains <= 6e-3 (bf16) / 8e-4 (fp16) rel-L2 on the BRANCH output (y - x: the part
the chain computes; the residual add itself is exact up to one final rounding), as for the per-layer LayerNorm-folded GEMMs.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from golden_util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bf16", "fp16"])
def ops(request):
    from unirestore_amd import ops as o
    o.set_dtype(request.param)
    yield o
    o.set_dtype("bf16")


def _mlp_ref(x, w1, b1, w2, b2, g, be, eps):
    h = F.layer_norm(x, (x.shape[-1],), g, be, eps)
    p = F.linear(h, w1, b1)
    a, gate = p.chunk(2, dim=-1)
    return x + F.linear(a * F.gelu(gate), w2, b2)


@pytest.mark.parametrize("rows", [128, 1024])
def test_ff_geglu_fused(ops, rows):
    from unirestore_amd import chain
    dt = ops.act_dtype()
    c, hid = 320, 1280
    gen = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, c, generator=gen) * 1.5 + 0.3).to(dt).float()
    w1 = torch.randn(2 * hid, c, generator=gen) / math.sqrt(c)
    b1 = torch.randn(2 * hid, generator=gen) * 0.1
    w2 = torch.randn(c, hid, generator=gen) / math.sqrt(hid)
    b2 = torch.randn(c, generator=gen) * 0.1
    g = 1 + 0.2 * torch.randn(c, generator=gen)
    be = 0.1 * torch.randn(c, generator=gen)
    ref = _mlp_ref(x.double(), w1.double(), b1.double(), w2.double(), b2.double(), g.double(), be.double(), 1e-5)
    st = chain.pack_mlp(w1, b1, w2, b2, g, be, "cuda")
    assert st.numel() == 3 * (hid // 64) * chain.TILE
    y = chain.ff_geglu_fused(x.to(dt).cuda(), st, hid, 1e-5).float().cpu()
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    assert rel_l2(y - x, ref.float() - x) < tol
    assert rel_l2(y, ref.float()) < tol
    # asymmetric check: a permuted-row / transposed-operand bug cannot hide behind symmetric inputs
    y2 = chain.ff_geglu_fused(x.to(dt).cuda(), st, hid, 1e-5).float().cpu()
    assert torch.equal(y, y2)                                   # deterministic


def test_ff_geglu_fused_rejects_unsupported(ops):
    from unirestore_amd import chain
    x = torch.zeros(128, 640, dtype=ops.act_dtype(), device="cuda")
    with pytest.raises(NotImplementedError):
        chain.ff_geglu_fused(x, torch.zeros(chain.TILE * 240, dtype=torch.uint8, device="cuda"), 2560, 1e-5)
    x = torch.zeros(100, 320, dtype=ops.act_dtype(), device="cuda")
    with pytest.raises(ValueError):
        chain.ff_geglu_fused(x, torch.zeros(chain.TILE * 60, dtype=torch.uint8, device="cuda"), 1280, 1e-5)


def _blk(c, cross, seed):
    """random weights of one Transformer2DModel (C = 320, 5 heads) as plain fp32 tensors"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    w = {}
    w["gn_g"], w["gn_b"] = 1 + 0.2 * r(c), 0.1 * r(c)
    w["in_w"], w["in_b"] = r(c, c, sc=c ** -0.5), r(c, sc=0.1)
    for i in (1, 2, 3):
        w[f"ln{i}_g"], w[f"ln{i}_b"] = 1 + 0.2 * r(c), 0.1 * r(c)
    for n in ("q1", "k1", "v1", "q2"):
        w[n] = r(c, c, sc=c ** -0.5)
    w["k2"], w["v2"] = r(c, cross, sc=cross ** -0.5), r(c, cross, sc=cross ** -0.5)
    for n in ("o1", "o2", "out"):
        w[n + "_w"], w[n + "_b"] = r(c, c, sc=c ** -0.5), r(c, sc=0.1)
    w["ff1_w"], w["ff1_b"] = r(8 * c, c, sc=c ** -0.5), r(8 * c, sc=0.1)
    w["ff2_w"], w["ff2_b"] = r(c, 4 * c, sc=(4 * c) ** -0.5), r(c, sc=0.1)
    w["ctx"] = r(77, cross)
    return w


def _attn(q, k, v, heads):
    b, t, c = q.shape
    d = c // heads
    qh, kh, vh = (z.reshape(b, -1, heads, d).transpose(1, 2) for z in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (p @ vh).transpose(1, 2).reshape(b, t, c)


def test_transformer_head_fused(ops):
    from unirestore_amd import chain
    dt = ops.act_dtype()
    n, hw, c = 2, 256, 320
    w = _blk(c, 1024, 5)
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn(n, hw, c, generator=gen) * 2 + 0.5).to(dt).float()
    ab = torch.stack([1 + 0.3 * torch.randn(n, c, generator=gen), 0.2 * torch.randn(n, c, generator=gen)], 1)      # [N][2][C]
    xn = (x * ab[:, 0:1] + ab[:, 1:2]).double()
    h0 = F.linear(xn, w["in_w"].double(), w["in_b"].double())
    hl = F.layer_norm(h0, (c,), w["ln1_g"].double(), w["ln1_b"].double(), 1e-5)
    ref = {"h0": h0, "q": F.linear(hl, w["q1"].double()), "k": F.linear(hl, w["k1"].double()), "v": F.linear(hl, w["v1"].double())}
    st = chain.pack_head(w["in_w"], w["in_b"], w["q1"], w["k1"], w["v1"], w["ln1_g"], w["ln1_b"], "cuda")
    assert st.numel() == 20 * chain.TILE
    h0_, q_, k_, vt_ = chain.transformer_head_fused(x.to(dt).cuda(), ab.cuda().contiguous(), st, n, 1e-5)
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    assert rel_l2(h0_.float().cpu().reshape(n, hw, c), ref["h0"].float()) < tol / 2
    assert rel_l2(q_.float().cpu(), ref["q"].float()) < tol
    assert rel_l2(k_.float().cpu(), ref["k"].float()) < tol
    assert rel_l2(vt_.float().cpu().transpose(1, 2), ref["v"].float()) < tol


@pytest.mark.parametrize("gn", [True, False])
def test_transformer_tail_fused(ops, gn):
    from unirestore_amd import chain
    dt = ops.act_dtype()
    n, hw, c, heads = 2, 256, 320, 5
    w = _blk(c, 1024, 7)
    gen = torch.Generator().manual_seed(13)
    o1, h0, x = ((torch.randn(n, hw, c, generator=gen) * s).to(dt).float() for s in (1.0, 1.5, 2.0))
    D = lambda t: t.double()
    h1 = D(h0) + F.linear(D(o1), D(w["o1_w"]), D(w["o1_b"]))
    q2 = F.linear(F.layer_norm(h1, (c,), D(w["ln2_g"]), D(w["ln2_b"]), 1e-5), D(w["q2"]))
    kc, vc = F.linear(D(w["ctx"]), D(w["k2"])), F.linear(D(w["ctx"]), D(w["v2"]))
    o2 = _attn(q2, kc[None].expand(n, -1, -1), vc[None].expand(n, -1, -1), heads)
    h2 = h1 + F.linear(o2, D(w["o2_w"]), D(w["o2_b"]))
    p = F.linear(F.layer_norm(h2, (c,), D(w["ln3_g"]), D(w["ln3_b"]), 1e-5), D(w["ff1_w"]), D(w["ff1_b"]))
    a, gate = p.chunk(2, dim=-1)
    h3 = h2 + F.linear(a * F.gelu(gate), D(w["ff2_w"]), D(w["ff2_b"]))
    ref = D(x) + F.linear(h3, D(w["out_w"]), D(w["out_b"]))
    st = chain.pack_tail(w["o1_w"], w["o1_b"], w["q2"], w["ln2_g"], w["ln2_b"], w["k2"], w["v2"], w["ctx"], w["o2_w"], w["o2_b"],
                         w["ff1_w"], w["ff1_b"], w["ff2_w"], w["ff2_b"], w["ln3_g"], w["ln3_b"], w["out_w"], w["out_b"], heads, "cuda")
    assert st.numel() == (25 + 60) * chain.TILE
    y = chain.transformer_tail_fused(o1.to(dt).cuda(), h0.to(dt).cuda(), x.to(dt).cuda(), st, n, 4 * c, heads, 77, 1e-5, 1 / 8.0, gn=gn)
    yf = y.float().cpu().reshape(n, hw, c)
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    assert rel_l2(yf - x, ref.float() - x) < tol
    if gn:
        part, parts = y._gn
        assert parts == hw // 128 and tuple(part.shape) == (n, parts, c, 2)
        s = part.sum(1).cpu()
        assert torch.allclose(s[..., 0], yf.sum(1), rtol=1e-4, atol=1e-2)
        assert torch.allclose(s[..., 1], (yf * yf).sum(1), rtol=1e-4, atol=1e-2)
    y2 = chain.transformer_tail_fused(o1.to(dt).cuda(), h0.to(dt).cuda(), x.to(dt).cuda(), st, n, 4 * c, heads, 77, 1e-5, 1 / 8.0, gn=gn)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("hw", [128, 1024])
def test_csce_fused(ops, hw):
    """SC-Tuner adapter in one launch (scedit.py:24-38: s = x + proj(cond); out = tuner(s) + s) vs fp64 torch, and vs the three
    per-layer GEMMs of the same module."""
    from unirestore_amd import chain
    from unirestore_amd.modules import adapters, nn as unn
    dt = ops.act_dtype()
    n, c, cc = 2, 320, 256
    gen = torch.Generator().manual_seed(hw)
    x = (torch.randn(n, hw, c, generator=gen) * 1.5).to(dt).float()
    cond = (torch.randn(n, hw, cc, generator=gen)).to(dt).float()
    wp, w0, w2 = (torch.randn(c, k, generator=gen) / math.sqrt(k) for k in (cc, c, c))
    bp, b0, b2 = (torch.randn(c, generator=gen) * 0.1 for _ in range(3))
    D = lambda t: t.double()
    s = D(x) + F.linear(D(cond), D(wp), D(bp))
    ref = (F.linear(F.gelu(F.linear(s, D(w0), D(b0))), D(w2), D(b2)) + s).float()
    st = chain.pack_csce(wp, bp, w0, b0, w2, b2, "cuda")
    assert st.numel() == 14 * chain.TILE
    xg, cg = x.to(dt).cuda().view(n, hw // 32, 32, c), cond.to(dt).cuda().view(n, hw // 32, 32, cc)
    y = chain.csce_fused(xg, cg, st)
    yf = y.float().cpu().reshape(n, hw, c)
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    assert rel_l2(yf - x, ref - x) < tol
    part, parts = y._gn
    assert parts == hw // 128 and tuple(part.shape) == (n, parts, c, 2)
    sm = part.sum(1).cpu()
    assert torch.allclose(sm[..., 0], yf.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sm[..., 1], (yf * yf).sum(1), rtol=1e-4, atol=1e-2)
    assert torch.equal(y, chain.csce_fused(xg, cg, st))
    # the module takes the chain on its own and agrees with its per-layer path
    m = adapters.CSCEAdapter(c, c, cc)
    with torch.no_grad():
        for p_, v in ((m.proj.weight, wp[:, :, None, None]), (m.proj.bias, bp), (m.tuner["0"].weight, w0[:, :, None, None]),
                      (m.tuner["0"].bias, b0), (m.tuner["2"].weight, w2[:, :, None, None]), (m.tuner["2"].bias, b2)):
            p_.copy_(v)
    assert unn.CHAIN
    ya = m.run(xg, cg)
    assert torch.equal(ya, y)
    unn.CHAIN = False
    try:
        yb = m.run(xg, cg)
    finally:
        unn.CHAIN = True
    assert rel_l2(ya.float().cpu() - xg.float().cpu(), yb.float().cpu() - xg.float().cpu()) < 2 * tol


def test_csce_fused_rejects_unsupported(ops):
    from unirestore_amd import chain
    dt = ops.act_dtype()
    st = torch.zeros(14 * chain.TILE, dtype=torch.uint8, device="cuda")
    x = torch.zeros(1, 96, 1, 320, dtype=dt, device="cuda")             # 96 tokens per image: not a multiple of 128
    with pytest.raises(ValueError):
        chain.csce_fused(x, torch.zeros(1, 96, 1, 256, dtype=dt, device="cuda"), st)
    x = torch.zeros(1, 128, 1, 640, dtype=dt, device="cuda")
    with pytest.raises(NotImplementedError):
        chain.csce_fused(x, torch.zeros(1, 128, 1, 256, dtype=dt, device="cuda"), st)
