"""HIP kernels vs plain PyTorch fp32 on the same 16-bit-rounded inputs (per-operator parity, -m gpu), run once per
16-bit type (bf16 and fp16).

Tolerances (stated): 16-bit outputs rel-L2 <= 3e-3 for bf16 / 4e-4 for fp16 (one rounding of the result is ~1.1e-3 /
1.4e-4 RMS); fp32 outputs rel-L2 <= 2e-4 (products of 16-bit values are exact in fp32; only summation order differs).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from golden_util import rel_l2

pytestmark = pytest.mark.gpu
TOL_BF16, TOL_F32 = 3e-3, 2e-4          # TOL_BF16 is re-bound per dtype by the `ops` fixture (3e-3 bf16, 4e-4 fp16)
DT = torch.bfloat16


@pytest.fixture(params=["bf16", "fp16"])
def ops(request):
    """The op front end with the compute dtype under test selected; module globals follow it."""
    global DT, TOL_BF16
    from unirestore_amd import ops as o
    DT = o.set_dtype(request.param)
    TOL_BF16 = 3e-3 if request.param == "bf16" else 4e-4
    yield o
    o.set_dtype("bf16")


def _rb(t):  # round to the 16-bit type under test and back
    return t.to(DT).float()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DT).cuda()


def _nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad", [
    (2, 64, 128, 16, 16, 3, 1, 1), (1, 320, 320, 32, 32, 3, 1, 1), (2, 40, 72, 9, 13, 3, 1, 1),
    (2, 64, 64, 16, 16, 3, 2, 1), (1, 128, 256, 24, 8, 1, 1, 0), (2, 8, 320, 16, 16, 3, 1, 1),
    (1, 128, 4, 32, 32, 3, 1, 1), (2, 1280, 1280, 8, 8, 3, 1, 1), (8, 640, 320, 8, 8, 1, 1, 0),
])
def test_conv_basic(ops, n, cin, cout, h, w, k, stride, pad):
    g = _gen(cin * 7 + cout)
    x = _rb(torch.randn(n, cin, h, w, generator=g))
    wt = _rb(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, stride=stride, padding=pad)
    pc = ops.pack_conv(wt, b, "cuda")
    y = ops.conv(_nhwc(x), pc, stride=stride, pad=(pad, pad))
    assert rel_l2(_nchw(y)[:, :cout], ref) < TOL_BF16
    y32 = ops.conv(_nhwc(x), pc, stride=stride, pad=(pad, pad), out_f32=True)
    assert rel_l2(_nchw(y32)[:, :cout], ref) < TOL_F32


def test_conv_asym_pad_stride2(ops):
    """VAE-encoder Downsample2D: F.pad(x,(0,1,0,1)) then 3x3 stride 2, padding 0."""
    g = _gen(3)
    x = _rb(torch.randn(2, 64, 16, 16, generator=g)); wt = _rb(torch.randn(64, 64, 3, 3, generator=g) / 24); b = torch.randn(64, generator=g)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, b, stride=2)
    y = ops.conv(_nhwc(x), ops.pack_conv(wt, b, "cuda"), stride=2, pad=(0, 0), out_hw=(8, 8))
    assert rel_l2(_nchw(y), ref) < TOL_BF16


def test_conv_concat_upsample_residual_act(ops):
    g = _gen(4)
    xa = _rb(torch.randn(2, 64, 8, 8, generator=g)); xb = _rb(torch.randn(2, 32, 8, 8, generator=g))
    wt = _rb(torch.randn(96, 96, 3, 3, generator=g) / 30); b = torch.randn(96, generator=g)
    res = _rb(torch.randn(2, 96, 8, 8, generator=g))
    pc = ops.pack_conv(wt, b, "cuda")
    ref = F.conv2d(torch.cat([xa, xb], 1), wt, b, padding=1) + res
    y = ops.conv(_nhwc(xa), pc, x2=_nhwc(xb), residual=_nhwc(res))
    assert rel_l2(_nchw(y), ref) < TOL_BF16
    x96 = torch.cat([xa, xb], 1)
    ref = F.conv2d(F.interpolate(x96, scale_factor=2.0, mode="nearest"), wt, b, padding=1)
    y = ops.conv(_nhwc(x96), pc, upsample=True)
    assert rel_l2(_nchw(y), ref) < TOL_BF16
    for act, fn in ((ops.UR_ACT_SILU, F.silu), (ops.UR_ACT_GELU, F.gelu)):
        y = ops.conv(_nhwc(x96), pc, act=act)
        assert rel_l2(_nchw(y), fn(F.conv2d(x96, wt, b, padding=1))) < TOL_BF16


def test_linear_geglu_gate_splitk(ops):
    g = _gen(5)
    x = _rb(torch.randn(2, 96, 128, generator=g))
    wt = _rb(torch.randn(1024, 128, generator=g) / 11); b = torch.randn(1024, generator=g)
    a, gt = F.linear(x, wt, b).chunk(2, -1)
    y = ops.linear(x.to(DT).cuda(), ops.pack_conv(wt, b, "cuda", pair=True), act=ops.UR_ACT_GEGLU)
    assert y.shape[-1] == 512 and rel_l2(y.float().cpu(), a * F.gelu(gt)) < TOL_BF16
    y = ops.linear(x.to(DT).cuda(), ops.pack_conv(wt, b, "cuda", pair=True), act=ops.UR_ACT_GATE)
    assert rel_l2(y.float().cpu(), a * gt) < TOL_BF16
    # small M, long K -> split-K path (with residual + pair act through the reduce kernel)
    x = _rb(torch.randn(64, 2048, generator=g)); wt = _rb(torch.randn(256, 2048, generator=g) / 45); b = torch.randn(256, generator=g)
    r = _rb(torch.randn(64, 256, generator=g))
    y = ops.linear(x.to(DT).cuda(), ops.pack_conv(wt, b, "cuda"), residual=r.to(DT).cuda())
    assert rel_l2(y.float().cpu(), F.linear(x, wt, b) + r) < TOL_BF16
    a, gt = F.linear(x, wt, b).chunk(2, -1)
    y = ops.linear(x.to(DT).cuda(), ops.pack_conv(wt, b, "cuda", pair=True), act=ops.UR_ACT_GEGLU)
    assert rel_l2(y.float().cpu(), a * F.gelu(gt)) < TOL_BF16


def test_conv_grouped_colsum_transposed(ops):
    g = _gen(6)
    x = _rb(torch.randn(2, 128, 8, 8, generator=g)); wt = _rb(torch.randn(128, 8, 3, 3, generator=g) / 8.5); b = torch.randn(128, generator=g)
    ref = F.gelu(F.conv2d(x, wt, b, padding=1, groups=16))
    y = ops.conv(_nhwc(x), ops.pack_conv(wt, b, "cuda", groups=16), act=ops.UR_ACT_GELU)
    assert rel_l2(_nchw(y), ref) < TOL_BF16
    # column sums (fused global average pool), no spatial output written
    x = _rb(torch.randn(2, 64, 16, 16, generator=g)); wt = _rb(torch.randn(96, 64, 3, 3, generator=g) / 24); b = torch.randn(96, generator=g)
    pc = ops.pack_conv(wt, b, "cuda")
    assert ops.conv_plan(_nhwc(x), pc, gn=True, store=False).gn_fused
    part, nparts = ops.conv(_nhwc(x), pc, gn=True, store=False)          # statistics-only launch: nothing else is written
    cs = ops.gn_finalize_planes(part, nparts, 256)
    assert rel_l2(cs.cpu(), F.conv2d(x, wt, b, padding=1).mean((2, 3))) < 1e-3
    # transposed second output (V^T for attention)
    x = _rb(torch.randn(2, 64, 128, generator=g)); wt = _rb(torch.randn(384, 128, generator=g) / 11)
    vt = torch.zeros(2, 128, 64, dtype=DT, device="cuda")
    y = ops.linear(x.to(DT).cuda(), ops.pack_conv(wt, None, "cuda"), yt=vt, n_split=256, t_rows=64)
    ref = F.linear(x, wt)
    assert rel_l2(y.float().cpu()[..., :256], ref[..., :256]) < TOL_BF16
    assert rel_l2(vt.float().cpu(), ref[..., 256:].transpose(1, 2)) < TOL_BF16


@pytest.mark.parametrize("b,t,c", [(1, 64, 1280), (2, 64, 1280), (8, 64, 1280), (1, 256, 1280), (2, 64, 640)])
def test_layernorm_folded_qkv_writes_v_transposed(ops, b, t, c):
    """Fused, LayerNorm-folded QKV GEMM with V written transposed, at the small row counts of the 8x8 / 16x16 levels at B = 1-2 -
    the launches that go through split-K + the row-wise reduce.  Round 5 found that reduce pass ignoring `yt`: V^T stayed
    uninitialised for M <= 128 rows (B = 1-2 at the 8x8 level), the dtype-independent 1.6e-3 of eps that kept fp16 zt at 8.1e-4."""
    g = _gen(b * t + c)
    x0 = _rb(torch.randn(b * t, c, generator=g)); w0 = _rb(torch.randn(c, c, generator=g) / math.sqrt(c))
    h2 = ops.linear(x0.to(DT).cuda(), ops.pack_conv(w0, None, "cuda"), rows=True)             # producer: leaves the row sums
    w = _rb(torch.randn(3 * c, c, generator=g) / math.sqrt(c))
    ga, be = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    vt = torch.full((b, c, t), float("nan"), dtype=DT, device="cuda")                       # NaN where the kernel does not write
    qk = ops.linear(ops.carry(h2, h2.view(b, t, c)), ops.pack_linear_ln(w, None, ga, be, 1e-5, "cuda"), ln_stats=ops.ln_of(h2), yt=vt, n_split=2 * c, t_rows=t)
    ref = F.linear(F.layer_norm(h2.float().cpu(), (c,), ga, be, 1e-5), w).view(b, t, 3 * c)
    assert rel_l2(qk.float().cpu()[..., :2 * c], ref[..., :2 * c]) < 2 * TOL_BF16
    assert bool(torch.isfinite(vt.float()).all()) and rel_l2(vt.float().cpu(), ref[..., 2 * c:].transpose(1, 2)) < 2 * TOL_BF16


@pytest.mark.parametrize("n,h,w,cin,cout,f32", [(8, 64, 64, 320, 4, True), (2, 64, 96, 128, 3, True), (8, 64, 64, 512, 8, False), (4, 32, 64, 64, 32, False)])
def test_conv_thin_output_halo(ops, n, h, w, cin, cout, f32):
    """conv_out layers (<= 32 output channels, fp32 or 16-bit): the thin halo tile (round 5) against torch, incl. the zero border."""
    g = _gen(n * h + cin + cout)
    x = _rb(torch.randn(n, cin, h, w, generator=g)); wt = _rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)); b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    y = ops.conv(_nhwc(x), ops.pack_conv(wt, b, "cuda"), out_f32=f32)
    got = y.float().cpu()[..., :cout].permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < (TOL_F32 if f32 else TOL_BF16)
    if y.shape[-1] > cout:
        assert float(y.float().cpu()[..., cout:].abs().max()) == 0.0          # padded output channels stay exactly zero


@pytest.mark.parametrize("m,k,n", [(64, 1280, 1280), (128, 1280, 3840), (512, 2560, 1280), (256, 640, 1920), (2048, 5120, 1280), (192, 1280, 640), (64, 1280, 3840),
                                   (64, 512, 1536), (512, 2560, 3840)])
@pytest.mark.parametrize("feat", ["plain", "res", "res+rows", "ln", "ln+res+rows", "ln+yt", "yt", "gn+res", "ln+gn", "gelu", "ln+geglu"])
def test_linear_feature_matrix_small_m(ops, m, k, n, feat):
    """Every epilogue feature combination the model uses, at the SMALL row counts (B = 1-2 at the 8x8 / 16x16 levels, long K) where the
    dispatcher goes through split-K and its reduce kernels - the corner that hid the round-5 V^T bug.  Against fp64 torch."""
    g = _gen(m + k + n + len(feat))
    f = set(feat.split("+"))
    if "yt" in f and n % 12:
        pytest.skip("transposed V needs N = 3 x C")
    pair = "geglu" in f
    x0 = _rb(torch.randn(m, 96, generator=g)); w0 = _rb(torch.randn(k, 96, generator=g) / math.sqrt(96))
    x = ops.linear(x0.to(DT).cuda(), ops.pack_conv(w0, None, "cuda"), rows="ln" in f)          # producer (leaves row sums when a LayerNorm consumer follows)
    xs = x.double().cpu()
    w = _rb(torch.randn(n, k, generator=g) / math.sqrt(k)); b = torch.randn(n, generator=g)
    kw = {}
    if "ln" in f:
        ga, be = 1 + 0.1 * torch.randn(k, generator=g), 0.1 * torch.randn(k, generator=g)
        pc = ops.pack_linear_ln(w, b, ga, be, 1e-5, "cuda", pair=pair)
        kw["ln_stats"] = ops.ln_of(x)
        ref = F.linear(F.layer_norm(xs, (k,), ga.double(), be.double(), 1e-5), w.double(), b.double())
    else:
        pc = ops.pack_conv(w, b, "cuda", pair=pair)
        ref = F.linear(xs, w.double(), b.double())
    if "gelu" in f:
        kw["act"] = ops.UR_ACT_GELU; ref = F.gelu(ref)
    if pair:
        kw["act"] = ops.UR_ACT_GEGLU
        a, gt = ref.chunk(2, -1); ref = a * F.gelu(gt)
    nout = ref.shape[1]
    if "res" in f:
        r = _rb(torch.randn(m, nout, generator=g)); kw["residual"] = r.to(DT).cuda(); ref = ref + r.double()
    if "rows" in f:
        kw["rows"] = True
    if "gn" in f:
        kw.update(gn=True, gn_hw=(1, m))
    vt = None
    if "yt" in f:
        c = n // 3
        vt = torch.full((1, c, m), float("nan"), dtype=DT, device="cuda")
        kw.update(yt=vt, n_split=2 * c, t_rows=m)
    y = ops.linear(x.view(1, m, k) if vt is not None else x, pc, **kw)
    yy = y.reshape(m, -1).double().cpu()
    tol = 2 * TOL_BF16 if "ln" in f else TOL_BF16
    ncmp = 2 * (n // 3) if vt is not None else nout
    assert rel_l2(yy[:, :ncmp], ref[:, :ncmp]) < tol
    if vt is not None:
        assert bool(torch.isfinite(vt.float()).all()) and rel_l2(vt[0].double().cpu(), ref[:, ncmp:].t()) < tol
    if "rows" in f:
        st, parts = ops.ln_of(y)
        ss = st.view(parts, m, 2).double().sum(0).cpu()
        assert rel_l2(ss[:, 0], yy.sum(1)) < 1e-5 and rel_l2(ss[:, 1], (yy ** 2).sum(1)) < 1e-5
    if "gn" in f:
        plane, parts = ops.gn_of(y)
        s2 = plane.double().sum(1)[0].cpu()
        assert rel_l2(s2[:, 0], yy.sum(0)) < 1e-4 and rel_l2(s2[:, 1], (yy ** 2).sum(0)) < 1e-5


@pytest.mark.parametrize("n,hw,cin,cout,cat", [(1, 8, 1280, 1280, 0), (4, 8, 1280, 1280, 1280), (2, 16, 1280, 1280, 0), (1, 16, 640, 1280, 640), (1, 32, 640, 640, 0),
                                               (2, 32, 320, 640, 320), (1, 64, 320, 320, 0)])
@pytest.mark.parametrize("feat", ["plain", "res+gn", "silu", "gn", "rowbias+gn"])
def test_conv3x3_feature_matrix_small_batch(ops, n, hw, cin, cout, cat, feat):
    """3x3 convs at B = 1-4 on the UNet's map sizes (whole-image halo tiles, channel-chunk split-K + the GroupNorm reduce pass, virtual
    concat) with the epilogue features the resnets use, against fp64 torch: the small-batch corner the B = 8 bench never visits."""
    g = _gen(n * hw + cin + cout + len(feat))
    f = set(feat.split("+"))
    x = _rb(torch.randn(n, cin, hw, hw, generator=g))
    x2 = _rb(torch.randn(n, cat, hw, hw, generator=g)) if cat else None
    wt = _rb(torch.randn(cout, cin + cat, 3, 3, generator=g) / math.sqrt(9 * (cin + cat))); b = torch.randn(cout, generator=g)
    xin = torch.cat([x, x2], 1) if cat else x
    ref = F.conv2d(xin.double(), wt.double(), None, padding=1)
    kw = {}
    if "rowbias" in f:                                       # per-image bias rows (time embeddings of a schedule-batched pass)
        rows = torch.randn(n, cout, generator=g)
        kw["bias"] = rows.cuda()
        ref = ref + rows.double()[:, :, None, None]
    else:
        ref = ref + b.double()[None, :, None, None]
    if "silu" in f:
        kw["act"] = ops.UR_ACT_SILU; ref = F.silu(ref)
    if "res" in f:
        r = _rb(torch.randn(n, cout, hw, hw, generator=g)); kw["residual"] = _nhwc(r); ref = ref + r.double()
    if "gn" in f:
        kw["gn"] = True
    y = ops.conv(_nhwc(x), ops.pack_conv(wt, b, "cuda", c1=cin if cat else None), x2=_nhwc(x2) if cat else None, **kw)
    got = _nchw(y).double()
    assert rel_l2(got, ref) < TOL_BF16
    if "gn" in f:
        plane, parts = ops.gn_of(y)
        s2 = plane.double().sum(1).cpu()                      # [n, cout, 2]
        assert rel_l2(s2[..., 0], got.sum((2, 3))) < 1e-4 and rel_l2(s2[..., 1], (got ** 2).sum((2, 3))) < 1e-5


@pytest.mark.parametrize("n,cin,cout,cat", [(8, 1280, 1280, 0), (8, 1280, 1280, 1280), (1, 512, 128, 0), (3, 768, 256, 256), (16, 512, 1280, 0), (5, 1280, 640, 0)])
@pytest.mark.parametrize("feat", ["plain", "res+gn", "rowbias+gn+silu"])
def test_conv3x3_weight_stream_8x8(ops, n, cin, cout, cat, feat):
    """The weight-streaming kernel of the 8 x 8 maps (csrc/conv_wstream.hip: fragment-major weights straight to registers, one chunk per
    wave, in-workgroup exchange, nchunk / 4 partial planes + the reduce pass) at the bench's B = 8 and at odd / small / large image counts
    (a half-empty image pair, 1 - 16 images), single source and virtual concat, against fp64 torch."""
    g = _gen(n * 13 + cin + cout + cat + len(feat))
    f = set(feat.split("+"))
    x = _rb(torch.randn(n, cin, 8, 8, generator=g))
    x2 = _rb(torch.randn(n, cat, 8, 8, generator=g)) if cat else None
    wt = _rb(torch.randn(cout, cin + cat, 3, 3, generator=g) / math.sqrt(9 * (cin + cat))); b = torch.randn(cout, generator=g)
    xin = torch.cat([x, x2], 1) if cat else x
    ref = F.conv2d(xin.double(), wt.double(), None, padding=1)
    kw = {}
    if "rowbias" in f:
        rows = torch.randn(n, cout, generator=g)
        kw["bias"] = rows.cuda()
        ref = ref + rows.double()[:, :, None, None]
    else:
        ref = ref + b.double()[None, :, None, None]
    if "silu" in f:
        kw["act"] = ops.UR_ACT_SILU; ref = F.silu(ref)
    if "res" in f:
        r = _rb(torch.randn(n, cout, 8, 8, generator=g)); kw["residual"] = _nhwc(r); ref = ref + r.double()
    if "gn" in f:
        kw["gn"] = True
    pc = ops.pack_conv(wt, b, "cuda", c1=cin if cat else None)
    y = ops.conv(_nhwc(x), pc, x2=_nhwc(x2) if cat else None, **kw)
    assert pc.w_frag is not None, "this shape must take the weight-streaming kernel"
    y2 = ops.conv(_nhwc(x), pc, x2=_nhwc(x2) if cat else None, **kw)
    assert torch.equal(y, y2)                                  # fixed-order exchange + reduce: bit-deterministic
    got = _nchw(y).double()
    assert rel_l2(got, ref) < TOL_BF16
    if "gn" in f:
        plane, parts = ops.gn_of(y)
        s2 = plane.double().sum(1).cpu()
        assert rel_l2(s2[..., 0], got.sum((2, 3))) < 1e-4 and rel_l2(s2[..., 1], (got ** 2).sum((2, 3))) < 1e-5


def test_bmm_nt(ops):
    g = _gen(7)
    a = _rb(torch.randn(3, 100, 64, generator=g)); b = _rb(torch.randn(3, 72, 64, generator=g))
    y = ops.bmm_nt(a.to(DT).cuda(), b.to(DT).cuda(), out_f32=True, out_scale=0.125)
    assert rel_l2(y.cpu(), a @ b.transpose(1, 2) * 0.125) < TOL_F32


@pytest.mark.parametrize("n,c,h,w,groups,silu", [(2, 320, 16, 16, 32, True), (1, 128, 32, 24, 32, False), (2, 64, 7, 9, 16, True),
                                                 (2, 2560, 4, 4, 32, True), (2, 64, 8, 8, 64, False)])
def test_groupnorm(ops, n, c, h, w, groups, silu):
    g = _gen(c)
    x = _rb(torch.randn(n, c, h, w, generator=g) * 2 + 0.5)
    inst = groups == c
    ga, be = (None, None) if inst else (torch.randn(c, generator=g), torch.randn(c, generator=g))
    ref = F.instance_norm(x, eps=1e-5) if inst else F.group_norm(x, groups, ga, be, eps=1e-5)
    ref = F.silu(ref) if silu else ref
    y = ops.group_norm(_nhwc(x), None if inst else ga.cuda(), None if inst else be.cuda(), groups, 1e-5, silu)
    assert rel_l2(_nchw(y), ref) < TOL_BF16


@pytest.mark.parametrize("rows,c", [(300, 320), (64, 1280), (1000, 64), (17, 640)])
def test_layernorm(ops, rows, c):
    g = _gen(rows)
    x = _rb(torch.randn(rows, c, generator=g) * 3 + 1); ga = torch.randn(c, generator=g); be = torch.randn(c, generator=g)
    y = ops.layer_norm(x.to(DT).cuda(), ga.cuda(), be.cuda(), 1e-5)
    assert rel_l2(y.float().cpu(), F.layer_norm(x, (c,), ga, be, 1e-5)) < TOL_BF16


def test_softmax_rows(ops):
    s = torch.randn(37, 333, generator=_gen(1)) * 4
    p = ops.softmax_rows(s.cuda())
    assert p.shape[-1] == 336 and float(p[:, 333:].abs().sum()) == 0
    assert rel_l2(p.float().cpu()[:, :333], torch.softmax(s, -1)) < TOL_BF16


@pytest.mark.parametrize("b,heads,d,tq,tk", [(2, 2, 64, 256, 256), (1, 5, 64, 1024, 77), (2, 4, 128, 64, 64), (1, 1, 64, 100, 200),
                                              (2, 4, 128, 256, 77), (2, 1, 512, 256, 256), (1, 1, 512, 100, 200), (1, 2, 512, 64, 77),
                                              (1, 1, 512, 1024, 1024)])
def test_attention(ops, b, heads, d, tq, tk):
    g = _gen(tq + tk + d)
    c = heads * d
    q = _rb(torch.randn(b, tq, c, generator=g)); k = _rb(torch.randn(b, tk, c, generator=g)); v = _rb(torch.randn(b, tk, c, generator=g))
    qh, kh, vh = (t.view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), -1) @ vh).transpose(1, 2).reshape(b, tq, c)
    ldvt = (tk + 7) // 8 * 8
    vt = torch.zeros(b, c, ldvt, dtype=DT); vt[:, :, :tk] = v.transpose(1, 2).to(DT)
    o = ops.attention(q.to(DT).cuda(), k.to(DT).cuda(), vt.cuda(), heads, d, tq, tk, 1 / math.sqrt(d),
                      ldq=c, ldk=c, bs_q=tq * c, bs_k=tk * c, bs_vt=c * ldvt, batch=b)
    assert rel_l2(o.float().cpu(), ref) < 2 * TOL_BF16   # P is rounded to 16 bits before PV (as SDPA's 16-bit path does)


@pytest.mark.parametrize("d", [64, 512])
def test_attention_softmax_spike(ops, d):
    """Force a large running-max jump mid-stream (online-softmax rescale path)."""
    g = _gen(9)
    b, heads, t = 1, 1, 256
    q = _rb(torch.randn(b, t, d, generator=g)); k = _rb(torch.randn(b, t, d, generator=g)); v = _rb(torch.randn(b, t, d, generator=g))
    k[0, 200] = q[0, 5] * 8
    ref = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), -1) @ v
    vt = v.transpose(1, 2).contiguous().to(DT)
    o = ops.attention(q.to(DT).cuda(), k.to(DT).cuda(), vt.cuda(), 1, d, t, t, 1 / math.sqrt(d), ldq=d, ldk=d,
                      bs_q=t * d, bs_k=t * d, bs_vt=d * t, batch=1)
    assert rel_l2(o.float().cpu(), ref) < 2 * TOL_BF16


def test_dwconv_pool_scale_misc(ops):
    g = _gen(10)
    x = _rb(torch.randn(2, 64, 9, 11, generator=g)); wt = torch.randn(64, 1, 3, 3, generator=g); b = torch.randn(64, generator=g)
    w9c = wt.view(64, 9).t().contiguous().cuda()
    ref = F.conv2d(x, wt, b, padding=1, groups=64)
    assert rel_l2(_nchw(ops.dwconv3x3(_nhwc(x), w9c, b.cuda())), ref) < TOL_BF16
    a, c = ref.chunk(2, 1)
    assert rel_l2(_nchw(ops.dwconv3x3(_nhwc(x), w9c, b.cuda(), gate=True)), a * c) < TOL_BF16
    assert rel_l2(ops.avgpool(_nhwc(x)).cpu(), x.mean((2, 3))) < 1e-5
    s = torch.randn(2, 64, generator=g); r = _rb(torch.randn(2, 64, 9, 11, generator=g))
    y = ops.scale_channels(_nhwc(x), s.cuda(), _nhwc(r))
    assert rel_l2(_nchw(y), x * s[:, :, None, None] + r) < TOL_BF16
    sc = torch.randn(64, generator=g)
    y = ops.axpy_channels(_nhwc(x), _nhwc(r), sc.cuda())
    assert rel_l2(_nchw(y), x + r * sc[None, :, None, None]) < TOL_BF16
    xm = torch.randn(5, 96, generator=g); w = torch.randn(40, 96, generator=g); bb = torch.randn(40, generator=g)
    assert rel_l2(ops.linear_f32(xm.cuda(), w.cuda(), bb.cuda(), ops.UR_ACT_SILU).cpu(), F.silu(F.linear(xm, w, bb))) < 1e-5
    wg = torch.randn(96, 24, generator=g)
    refg = torch.cat([F.linear(xm[:, i * 24:(i + 1) * 24], wg[i * 24:(i + 1) * 24]) for i in range(4)], 1)
    assert rel_l2(ops.linear_f32(xm.cuda(), wg.cuda(), None, groups=4).cpu(), refg) < 1e-5
    pooled = torch.randn(2, 3, 2 * 48, generator=g); cond = torch.randn(2, 2, 48, generator=g)
    f, i, cc = (pooled[:, j].view(2, 2, 48) for j in range(3))
    ref = torch.softmax(f, -1) * cond + torch.softmax(i, -1) * torch.tanh(cc)
    assert rel_l2(ops.tfa_prompt_update(pooled.cuda(), cond.cuda()).cpu(), ref) < 1e-5


def test_boundary_kernels(ops):
    g = _gen(11)
    img = torch.rand(2, 3, 16, 24, generator=g)
    y = ops.nchw_to_nhwc(img.cuda(), image=True)
    assert y.shape == (2, 16, 24, 8) and float(y[..., 3:].abs().sum()) == 0
    assert rel_l2(_nchw(y)[:, :3], img * 2 - 1) < TOL_BF16
    back = ops.nhwc_to_nchw(y, c=3, mul=0.5, add=0.5)
    assert rel_l2(back.cpu(), img) < 4e-3
    mom = torch.randn(2, 8, 8, 8, generator=g); noise = torch.randn(2, 4, 8, 8, generator=g)
    z, zb = ops.vae_sample(mom.cuda(), noise.cuda(), 4, 0.18215)
    mean, lv = mom.permute(0, 3, 1, 2).chunk(2, 1)
    ref = (mean + torch.exp(0.5 * lv.clamp(-30, 20)) * noise) * 0.18215
    assert rel_l2(z.cpu().permute(0, 3, 1, 2)[:, :4], ref) < 1e-6 and float(z[..., 4:].abs().sum()) == 0
    zt, ztb = ops.add_noise(z, noise.cuda(), 4, 0.3, 0.9)
    assert rel_l2(zt.cpu().permute(0, 3, 1, 2)[:, :4], 0.3 * ref + 0.9 * noise) < 1e-6
    eps = torch.randn(2, 8, 8, 4, generator=g).cuda()
    before = zt.clone()
    ops.ddim_step_(zt, ztb, eps, 4, 1.7, -0.4)
    assert rel_l2(zt[..., :4].cpu(), (1.7 * before[..., :4] - 0.4 * eps).cpu()) < 1e-6
    assert rel_l2(ztb.float().cpu(), zt.cpu()) < TOL_BF16


def test_error_convention(ops):
    x = torch.zeros(1, 4, 4, 12, dtype=DT, device="cuda")
    pc = ops.pack_conv(torch.zeros(8, 16, 3, 3), None, "cuda")
    with pytest.raises(ValueError):               # UR_E_INVALID -> ValueError (Cin mismatch: 12 vs 16 channels)
        ops.conv(x, pc)
    with pytest.raises(NotImplementedError):      # UR_E_UNSUPPORTED: gate epilogue on 16 output channels (< 32-row a|g blocks)
        ops.pack_conv(torch.zeros(32, 16, 1, 1), None, "cuda", pair=True)
    with pytest.raises(NotImplementedError):      # a statistics-only launch where the epilogue cannot produce them
        xs = torch.zeros(1, 3, 5, 16, dtype=DT, device="cuda")
        ops.conv(xs, ops.pack_conv(torch.zeros(8, 16, 3, 3), None, "cuda"), gn=True, store=False)


def test_groupnorm_virtual_concat(ops):
    """GroupNorm over cat([a, b]) with a group straddling the boundary (1280+640 -> 60 ch/group in the UNet)."""
    g = _gen(12)
    a = _rb(torch.randn(2, 128, 8, 8, generator=g)); b = _rb(torch.randn(2, 64, 8, 8, generator=g) * 2 + 1)
    ga, be = torch.randn(192, generator=g), torch.randn(192, generator=g)
    ref = F.silu(F.group_norm(torch.cat([a, b], 1), 32, ga, be, eps=1e-5))      # 6 channels / group
    y = ops.group_norm(_nhwc(a), ga.cuda(), be.cuda(), 32, 1e-5, True, x2=_nhwc(b))
    assert rel_l2(_nchw(y), ref) < TOL_BF16


def test_conv_per_image_bias(ops):
    g = _gen(13)
    x = _rb(torch.randn(3, 32, 8, 8, generator=g)); wt = _rb(torch.randn(64, 32, 3, 3, generator=g) / 17)
    bi = torch.randn(3, 64, generator=g)
    y = ops.conv(_nhwc(x), ops.pack_conv(wt, None, "cuda"), bias=bi.cuda())
    assert rel_l2(_nchw(y), F.conv2d(x, wt, padding=1) + bi[:, :, None, None]) < TOL_BF16


@pytest.mark.parametrize("n,cin,cout,h,w,k", [(2, 64, 128, 16, 16, 3), (8, 320, 320, 16, 16, 3), (2, 64, 64, 8, 8, 1), (1, 128, 256, 24, 8, 1),
                                               (2, 1280, 640, 16, 16, 3), (3, 640, 1280, 8, 8, 3),       # split-K reduce leaves the sums
                                               (2, 320, 320, 32, 32, 3), (1, 640, 640, 32, 64, 3)])       # chunk-split 8x32 halo patches
def test_conv_fused_groupnorm_stats(ops, n, cin, cout, h, w, k):
    """The conv epilogue (or its fallback pass) leaves per-image channel sums; GroupNorm consumes them without a stats pass."""
    g = _gen(cout + h)
    x = _rb(torch.randn(n, cin, h, w, generator=g)); wt = _rb(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g); ga, be = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    r = _rb(torch.randn(n, cout, h, w, generator=g))
    y = ops.conv(_nhwc(x), ops.pack_conv(wt, b, "cuda"), gn=True, residual=_nhwc(r))
    assert rel_l2(_nchw(y), F.conv2d(x, wt, b, padding=k // 2) + r) < TOL_BF16
    st = ops.gn_of(y)[0].double().sum(1).cpu()          # partial planes [N][P][C][2] -> [N][C][2]
    yf = _nchw(y).double()
    assert rel_l2(st[..., 0], yf.sum((2, 3))) < 1e-5 and rel_l2(st[..., 1], (yf * yf).sum((2, 3))) < 1e-5
    out = ops.group_norm(y, ga.cuda(), be.cuda(), 32, 1e-5, True)
    ref = F.silu(F.group_norm(_nchw(y), 32, ga, be, eps=1e-5))
    assert rel_l2(_nchw(out), ref) < TOL_BF16


@pytest.mark.parametrize("n,cin,cout,h,w,ups,res", [(2, 64, 128, 16, 32, False, False), (1, 320, 320, 32, 32, False, True),
                                                    (8, 128, 160, 8, 64, False, False), (2, 192, 256, 8, 16, True, True),
                                                    (3, 256, 128, 8, 8, True, False), (8, 640, 1280, 8, 8, True, True)])   # 8x8 -> 16x16: whole-image tile, chunk-split
def test_conv_halo_tile_path(ops, n, cin, cout, h, w, ups, res):
    """3x3 / stride 1 / pad 1 shapes that take the LDS halo-tile kernels (8x32 output patches; whole 16x16 images), incl. fused upsample."""
    g = _gen(cin + cout + h)
    x = _rb(torch.randn(n, cin, h, w, generator=g)); wt = _rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, wt, b, padding=1)
    r = _rb(torch.randn(ref.shape, generator=g)) if res else None
    y = ops.conv(_nhwc(x), ops.pack_conv(wt, b, "cuda"), upsample=ups, residual=None if r is None else _nhwc(r), gn=True)
    ref = ref + r if res else ref
    assert rel_l2(_nchw(y), ref) < TOL_BF16
    st = ops.gn_of(y)[0].double().sum(1).cpu()          # partial planes [N][P][C][2] -> [N][C][2]
    assert rel_l2(st[..., 0], _nchw(y).double().sum((2, 3))) < 1e-5


@pytest.mark.parametrize("rows,c,n,pair,k0", [
    (256, 320, 960, False, 96), (300, 64, 128, False, 96), (512, 320, 2560, True, 96), (128, 640, 640, False, 96),
    (5200, 320, 2560, True, 96),             # >= 200 tiles of 256 x 256 -> gemm_glds_kernel
    (512, 1280, 1280, False, 1024),          # few tiles, long K: producer AND consumer go through split-K + the row-wise reduce
    (200, 1280, 2560, True, 640), (2048, 1280, 1280, False, 96),
    (2048, 320, 10240, True, 96)])           # 256 x 320 gated tiles with the LayerNorm consumer
def test_layernorm_fused_into_gemm(ops, rows, c, n, pair, k0):
    """Producer GEMM leaves per-row sums; the consumer computes Linear(LayerNorm(x)) without a LayerNorm pass."""
    g = _gen(rows + n)
    x0 = _rb(torch.randn(rows, k0, generator=g)); w0 = _rb(torch.randn(c, k0, generator=g) / math.sqrt(k0)); r0 = _rb(torch.randn(rows, c, generator=g) * 2 + 0.5)
    x = ops.linear(x0.to(DT).cuda(), ops.pack_conv(w0, None, "cuda"), residual=r0.to(DT).cuda(), rows=True)
    xs = x.float().cpu()
    stt, parts = ops.ln_of(x)
    st = stt.view(parts, rows, 2).sum(0).cpu()
    assert rel_l2(st[:, 0], xs.double().sum(1)) < 1e-5 and rel_l2(st[:, 1], (xs.double() ** 2).sum(1)) < 1e-5
    ga, be = torch.randn(c, generator=g), torch.randn(c, generator=g)
    w = _rb(torch.randn(n, c, generator=g) / math.sqrt(c)); b = torch.randn(n, generator=g)
    ref = F.linear(F.layer_norm(xs, (c,), ga, be, 1e-5), w, b)
    pc = ops.pack_linear_ln(w, b, ga, be, 1e-5, "cuda", pair=pair)
    y = ops.linear(x, pc, ln_stats=ops.ln_of(x), act=ops.UR_ACT_GEGLU if pair else ops.UR_ACT_NONE)
    if pair:
        a, gt = ref.chunk(2, -1)
        ref = a * F.gelu(gt)
    assert rel_l2(y.float().cpu(), ref) < 2 * TOL_BF16      # gamma is folded into the 16-bit weights: one extra rounding of W*gamma


@pytest.mark.parametrize("m,k,n,act", [(5200, 320, 2560, "geglu"), (8192, 192, 1792, "gate"),
                                       (2048, 640, 10240, "geglu"), (2000, 320, 5120, "gate")])     # last two: 256 x 320 tiles
def test_linear_pair_act_256_tiles(ops, m, k, n, act):
    """Large-N pair-activation GEMMs take the 256 x 256 pure-GEMM tile kernel (ragged M, K tail < 64)."""
    g = _gen(m + n)
    x = _rb(torch.randn(m, k, generator=g)); wt = _rb(torch.randn(n, k, generator=g) / math.sqrt(k)); b = torch.randn(n, generator=g)
    a, gt = F.linear(x, wt, b).chunk(2, -1)
    ref = a * F.gelu(gt) if act == "geglu" else a * gt
    y = ops.linear(x.to(DT).cuda(), ops.pack_conv(wt, b, "cuda", pair=True),
                   act=ops.UR_ACT_GEGLU if act == "geglu" else ops.UR_ACT_GATE)
    assert y.shape == (m, n // 2) and rel_l2(y.float().cpu(), ref) < TOL_BF16


@pytest.mark.parametrize("h,w,rh,rw,ph,pw", [(37, 53, 90, 131, 6, 29), (64, 96, 40, 60, 24, 4), (48, 48, 48, 48, 16, 16), (33, 47, 33, 47, 0, 0)])
def test_image_resize_reflect_pad(ops, h, w, rh, rw, ph, pw):
    """unifie.py:124-134 as one HIP kernel: bicubic (align_corners=False, no antialias) + reflect pad + x*2-1 + NHWC bf16."""
    img = torch.rand(2, 3, h, w, generator=_gen(h * w))
    ref = F.interpolate(img, (rh, rw), mode="bicubic", align_corners=False, antialias=False) if (rh, rw) != (h, w) else img
    if ph or pw:
        ref = F.pad(ref, (0, pw, 0, ph), mode="reflect")
    ref = ref * 2 - 1
    y = ops.image_resize_pad(img.cuda(), rh, rw, ph, pw)
    assert y.shape == (2, rh + ph, rw + pw, 8) and float(y[..., 3:].float().abs().max()) == 0.0
    got = y[..., :3].float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max() < 8e-3 and rel_l2(got, ref) < TOL_BF16          # one bf16 rounding of values in [-1.3, 1.3]


@pytest.mark.parametrize("f32", [True, False])
@pytest.mark.parametrize("xh,xw,ch,cw,oh,ow", [(24, 40, 20, 33, 37, 53), (64, 64, 64, 50, 32, 25), (16, 24, 16, 24, 16, 24)])
def test_image_unpad_resize_quantize(ops, f32, xh, xw, ch, cw, oh, ow):
    """unifie.py:164-168 (+ the evaluator's 8-bit quantisation, eval_image_restoration.py:71) as one HIP kernel."""
    x = torch.randn(2, xh, xw, 8, generator=_gen(xh + ow)) * 0.6
    x = x if f32 else _rb(x)
    xd = (x if f32 else x.to(DT)).cuda()
    ref = (x[..., :3] * 0.5 + 0.5)[:, :ch, :cw].permute(0, 3, 1, 2)
    ref = F.interpolate(ref, (oh, ow), mode="bicubic", align_corners=False, antialias=False)
    got = ops.image_unpad_resize(xd, 3, (ch, cw), (oh, ow), mul=0.5, add=0.5).cpu()
    assert got.shape == ref.shape and (got - ref).abs().max() < 2e-5
    q = ops.image_unpad_resize(xd, 3, (ch, cw), (oh, ow), mul=0.5, add=0.5, quantize=True).cpu()
    qref = ref.mul(255).round().clamp(0, 255).div(255)
    d = (q - qref).abs()
    assert d.max() <= 1.0 / 255 + 1e-6 and float((d > 1e-6).float().mean()) < 2e-3     # a rounding tie may flip one code value


@pytest.mark.parametrize("n,c1,c2,cout,hw,img_bias", [(2, 1280, 0, 640, 16, False), (8, 256, 0, 128, 16, False), (3, 640, 320, 1280, 16, True),
                                                      (4, 1280, 0, 1280, 8, False), (8, 640, 640, 256, 8, True), (4, 256, 0, 128, 8, False)])
def test_conv_whole_image_halo_tiles(ops, n, c1, c2, cout, hw, img_bias):
    """16x16 / 8x8 maps: whole-image halo tiles (one image / four images per tile), split over channel chunks, with the
    virtual concat, residual, per-image bias rows and the GroupNorm sums coming out of the split-K reduce."""
    g = _gen(n * cout + hw + c2)
    cin = c1 + c2
    x = _rb(torch.randn(n, cin, hw, hw, generator=g)); wt = _rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = torch.randn(cout, generator=g); r = _rb(torch.randn(n, cout, hw, hw, generator=g))
    bi = torch.randn(n, cout, generator=g)
    xa = _nhwc(x[:, :c1]); xb = _nhwc(x[:, c1:]) if c2 else None
    pc = ops.pack_conv(wt, None if img_bias else b, "cuda")
    y = ops.conv(xa, pc, x2=xb, residual=_nhwc(r), gn=True, bias=bi.cuda() if img_bias else None)
    ref = F.conv2d(x, wt, None if img_bias else b, padding=1) + r + (bi[:, :, None, None] if img_bias else 0)
    assert rel_l2(_nchw(y), ref) < TOL_BF16
    st = ops.gn_of(y)[0].double().sum(1).cpu()          # partial planes [N][P][C][2] -> [N][C][2]
    yf = _nchw(y).double()
    assert rel_l2(st[..., 0], yf.sum((2, 3))) < 1e-5 and rel_l2(st[..., 1], (yf * yf).sum((2, 3))) < 1e-5


@pytest.mark.parametrize("n,hw,c1,c2,cout", [(8, 64, 640, 320, 320), (8, 32, 1280, 640, 640), (2, 16, 1280, 1280, 1280)])
def test_conv1x1_virtual_concat_shortcuts(ops, n, hw, c1, c2, cout):
    """The up path's 1x1 shortcuts read cat[h, skip] without materialising it (pure-GEMM LDS-DMA kernel and fallbacks)."""
    g = _gen(c1 + c2 + hw)
    x = _rb(torch.randn(n, c1 + c2, hw, hw, generator=g)); wt = _rb(torch.randn(cout, c1 + c2, 1, 1, generator=g) / math.sqrt(c1 + c2))
    b = torch.randn(cout, generator=g)
    y = ops.conv(_nhwc(x[:, :c1]), ops.pack_conv(wt, b, "cuda"), x2=_nhwc(x[:, c1:]))
    assert rel_l2(_nchw(y), F.conv2d(x, wt, b)) < TOL_BF16


@pytest.mark.parametrize("n,c1,c2,cout,h,w,ups", [(8, 128, 0, 256, 32, 64, False),       # 8x32 halo patches, 128 tiles
                                                  (4, 320, 320, 256, 32, 64, False),     # virtual concat
                                                  (8, 64, 0, 128, 16, 32, True),         # fused nearest-2x upsample
                                                  (8, 256, 0, 128, 32, 64, False),       # 64 tiles -> channel-chunk split + reduce pass
                                                  (8, 128, 64, 128, 24, 96, False),      # concat boundary inside the K range
                                                  (3, 256, 0, 128, 16, 16, False), (2, 640, 640, 256, 16, 16, False)])   # 16x16 whole-image tiles
def test_conv_groupnorm_prologue(ops, n, c1, c2, cout, h, w, ups):
    """GroupNorm apply + SiLU fused into the 3x3 conv's loader (ur_conv_desc.gn_ab: halo 8x32 patches and 16x16 whole-image
    tiles, virtual concat, fused upsample, chunk-split K): equal to applying the GroupNorm first, and to PyTorch."""
    g = _gen(c1 + c2 + cout + h)
    cin = c1 + c2
    x = _rb(torch.randn(n, cin, h, w, generator=g) * 1.5 + 0.3)
    wt = _rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)); b = torch.randn(cout, generator=g)
    ga, be = torch.randn(cin, generator=g), torch.randn(cin, generator=g)
    xa, xb = _nhwc(x[:, :c1]), (_nhwc(x[:, c1:]) if c2 else None)
    pc = ops.pack_conv(wt, b, "cuda", c1=c1 if c2 else None)
    plan = ops.conv_plan(xa, pc, x2=xb, upsample=ups, gn=True, gn_ab=True)
    assert plan.prologue_ok, "shape should take a halo kernel"
    ab = ops.gn_finalize(xa, ga.cuda(), be.cuda(), 32, 1e-5, x2=xb)
    fused = ops.conv(xa, pc, x2=xb, upsample=ups, gn_ab=ab, gn_silu=True, gn=True)
    two_pass = ops.conv(ops.gn_apply(xa, ab, silu=True, x2=xb), pc, upsample=ups, gn=True)
    assert torch.equal(fused, two_pass)                               # same rounded operands, same summation order
    assert torch.equal(ops.gn_of(fused)[0], ops.gn_of(two_pass)[0])
    hn = F.silu(F.group_norm(x, 32, ga, be, eps=1e-5))
    ref = F.conv2d(F.interpolate(hn, scale_factor=2.0, mode="nearest") if ups else hn, wt, b, padding=1)
    assert rel_l2(_nchw(fused), ref) < 2 * TOL_BF16                   # the normalised operand is rounded to 16 bits once more
    nosilu = ops.conv(xa, pc, x2=xb, upsample=ups, gn_ab=ab, gn_silu=False)
    assert torch.equal(nosilu, ops.conv(ops.gn_apply(xa, ab, silu=False, x2=xb), pc, upsample=ups))
    if not ops.conv_plan(_nhwc(x[:, :8, :5, :7]), ops.pack_conv(wt[:, :8], b, "cuda"), gn_ab=True).prologue_ok:
        with pytest.raises(NotImplementedError):                      # UR_E_UNSUPPORTED where the launch cannot honour gn_ab
            ops.conv(_nhwc(x[:, :8, :5, :7]), ops.pack_conv(wt[:, :8], b, "cuda"), gn_ab=ab, gn_silu=True)


@pytest.mark.parametrize("m,k,n", [(4100, 200, 264), (8192, 72, 640), (1000, 328, 1288)])
def test_linear_k_tail_and_ragged_tiles(ops, m, k, n):
    """1x1 / Linear shapes whose K is not a multiple of the 64-deep K tile and whose M / N do not fill the last tiles: the buffer-
    descriptor loaders return zeros for rows past M / Cout (range check) and for the K tail (per-lane select)."""
    g = _gen(m + k + n)
    x = _rb(torch.randn(m, k, generator=g)); w = _rb(torch.randn(n, k, generator=g) / math.sqrt(k)); b = torch.randn(n, generator=g)
    r = _rb(torch.randn(m, n, generator=g))
    y = ops.linear(x.to(DT).cuda(), ops.pack_conv(w, b, "cuda"), residual=r.to(DT).cuda())
    assert rel_l2(y.float().cpu(), F.linear(x, w, b) + r) < TOL_BF16


_HALO_AB_SNIPPET = r"""
import sys, math, torch
sys.path.insert(0, {root!r})
from unirestore_amd import ops
ops.set_dtype({dt!r})
DT = ops.act_dtype()
outs = []
for (n, cin, cout, h, w, ups) in {cases!r}:
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g).to(DT); wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(DT).float()
    b = torch.randn(cout, generator=g)
    y = ops.conv(x.permute(0, 2, 3, 1).contiguous().cuda(), ops.pack_conv(wt, b, "cuda"), upsample=ups, gn=True)
    outs.append(y.cpu()); outs.append(ops.gn_of(y)[0].cpu())
torch.save(outs, {out!r})
"""


def test_conv_halo_wave_specialised_equals_self_loading_kernel(ops, tmp_path):
    """The wave-specialised halo conv (8 compute waves + 1 loader wave, tap-crossing fragment pipeline, buffer-descriptor DMA) and the
    kernel whose waves load for themselves (UR_HALO_NOWS=1, read once per process: second process) accumulate in the same order:
    outputs and GroupNorm partial planes must be BIT-identical, on full tiles, ragged Cout tiles, the fused upsample and both widths."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [(2, 128, 320, 16, 64, False), (1, 192, 200, 8, 32, False), (2, 64, 256, 8, 16, True), (1, 320, 160, 24, 32, False)]
    dt = "fp16" if DT == torch.float16 else "bf16"
    res = []
    for tag, env in (("ws", {}), ("nows", {"UR_HALO_NOWS": "1"})):
        out = str(tmp_path / f"{tag}.pt")
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", _HALO_AB_SNIPPET.format(root=root, dt=dt, cases=cases, out=out)], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(out))
    assert len(res[0]) == len(res[1]) == 2 * len(cases)
    for a, b in zip(*res):
        assert torch.equal(a, b)


_XCD_AB_SNIPPET = r"""
import sys, math, torch
sys.path.insert(0, {root!r})
from unirestore_amd import ops
ops.set_dtype({dt!r})
DT = ops.act_dtype()
outs = []
for (m, k, n, pair) in {lin!r}:
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g).to(DT); w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(DT).float(); b = torch.randn(n, generator=g)
    r = None if pair else torch.randn(m, n, generator=g).to(DT).cuda()
    y = ops.linear(x.cuda(), ops.pack_conv(w, b, "cuda", pair=pair), residual=r, act=ops.UR_ACT_GEGLU if pair else ops.UR_ACT_NONE)
    outs.append(y.cpu())
for (nimg, cin, cout, hw, ups) in {convs!r}:
    g = torch.Generator().manual_seed(cin + cout + hw)
    x = torch.randn(nimg, hw, hw, cin, generator=g).to(DT); wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(DT).float()
    y = ops.conv(x.cuda(), ops.pack_conv(wt, torch.randn(cout, generator=g), "cuda"), upsample=ups, gn=True)
    outs.append(y.cpu()); outs.append(ops.gn_of(y)[0].cpu())
torch.save(outs, {out!r})
"""


def test_xcd_tile_grids_change_nothing_but_the_placement(ops, tmp_path):
    """Round 6: which XCD computes which tile is chosen per launch (pick_xcd_grid: 4 x 2 / 2 x 4 / 1 x 8 XCD grids, the row-tile-fastest
    fallback on ragged grids, the weight-major 1-D grid of the whole-image halo conv).  The arithmetic of a tile does not depend on where it
    runs: outputs (and GroupNorm partial planes) must be BIT-identical to the rounds-1-5 map (UR_NOXCDGRID=1 UR_HIMG_NOWMAJOR=1, read once
    per process: second process) - on the shapes of the 16x16 / 8x8 levels the map was built for, a ragged grid, a multi-round grid."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lin = [(2048, 1280, 3840, False), (512, 1280, 1280, False), (2048, 1280, 10240, True), (512, 1280, 10240, True), (2048, 5120, 1280, False),
           (2048, 1280, 1280, False), (8192, 640, 640, False), (1000, 328, 1288, False), (4096, 640, 1920, False), (512, 2560, 1280, False)]
    convs = [(8, 1280, 1280, 16, False), (8, 640, 1280, 16, False), (3, 256, 384, 16, False), (8, 512, 1280, 8, True),
             (2, 320, 320, 64, False), (1, 128, 128, 128, False), (4, 512, 512, 8, False), (2, 640, 640, 32, False)]      # 8 x 32 halo tiles (+ chunk split), 8x8x4 tiles
    dt = "fp16" if DT == torch.float16 else "bf16"
    res = []
    # (the second process also runs the rounds-3-5 loader structures: ONE loader wave in the 8 x 32 halo conv, the self-loading whole-image
    #  kernel - the two-loader kernels of round 6 accumulate in the same order)
    for tag, env in (("grid", {}), ("old", {"UR_NOXCDGRID": "1", "UR_HIMG_NOWMAJOR": "1", "UR_HALO_LD1": "1", "UR_HIMG_NOWS": "1"})):
        out = str(tmp_path / f"{tag}.pt")
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", _XCD_AB_SNIPPET.format(root=root, dt=dt, lin=lin, convs=convs, out=out)], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(out))
    assert len(res[0]) == len(res[1]) == len(lin) + 2 * len(convs)
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b), i
