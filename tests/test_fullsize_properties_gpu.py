"""Size-independent properties at BASELINE.json's full layer sizes (-m gpu): the CPU oracle is too slow there, so the
kernels are checked through identities the domain offers - linearity and translation equivariance of the convolution,
softmax rows that sum to one, key-permutation invariance of attention, zero-mean / unit-variance GroupNorm groups, image
independence of the whole model (no batch statistics anywhere).  Tolerances are bf16-rounding level."""
import math

import pytest
import torch

from golden_util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from unirestore_amd import ops as o
    o.set_dtype("bf16")
    return o


def _bf(t):
    return t.to(torch.bfloat16).cuda()


@pytest.mark.parametrize("b,hw,cin,cout", [(8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280), (8, 8, 1280, 1280), (1, 128, 320, 320)])
def test_conv3x3_linearity_and_translation(ops, b, hw, cin, cout):
    """UNet conv shapes at batch 8 (halo / whole-image-halo / split paths) and the 1024x1024 configuration's 128x128 level."""
    g = torch.Generator().manual_seed(hw + cin)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    pc = ops.pack_conv(w, None, "cuda")
    x1, x2 = _bf(torch.randn(b, hw, hw, cin, generator=g)), _bf(torch.randn(b, hw, hw, cin, generator=g))
    y1, y2 = ops.conv(x1, pc).float(), ops.conv(x2, pc).float()
    y12 = ops.conv((x1.float() * 0.5 + x2.float() * 0.25).to(torch.bfloat16), pc).float()
    assert rel_l2(y12.cpu(), (0.5 * y1 + 0.25 * y2).cpu()) < 6e-3                       # three bf16 roundings
    # translation: shifting the input two pixels along W shifts the output (away from the zero-padded border)
    xs = torch.zeros_like(x1)
    xs[:, :, 2:] = x1[:, :, :-2]
    ys = ops.conv(xs, pc).float()
    # (the last column sees the zero padding instead of x1's column W-1: excluded)
    assert rel_l2(ys[:, :, 2:-1].cpu(), y1[:, :, :-3].cpu()) < 1e-6                     # same products, same order: exact
    assert bool(torch.isfinite(y12).all())


@pytest.mark.parametrize("m,k,n", [(32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280), (32768, 1280, 320)])
def test_gemm_linearity(ops, m, k, n):
    g = torch.Generator().manual_seed(m + n)
    pc = ops.pack_conv(torch.randn(n, k, 1, 1, generator=g) / math.sqrt(k), None, "cuda")
    x1, x2 = _bf(torch.randn(m, k, generator=g)), _bf(torch.randn(m, k, generator=g))
    y = ops.linear((x1.float() - x2.float()).to(torch.bfloat16), pc).float()
    assert rel_l2(y.cpu(), (ops.linear(x1, pc).float() - ops.linear(x2, pc).float()).cpu()) < 6e-3


@pytest.mark.parametrize("t,heads", [(4096, 5), (1024, 10), (16384, 5)])
def test_attention_rows_sum_to_one_and_key_permutation(ops, t, heads):
    """V = 1 returns 1 (softmax rows sum to one); permuting the keys (and V columns alike) leaves the output unchanged.
    T = 16384 is the 1024x1024 configuration's self-attention length."""
    b, d = (2 if t > 4096 else 8), 64
    c = heads * d
    g = torch.Generator().manual_seed(t)
    qkv = _bf(torch.randn(b, t, 3 * c, generator=g))
    kw = dict(ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=b)
    ones = torch.ones(b, c, t, dtype=torch.bfloat16, device="cuda")
    o = ops.attention(qkv, qkv[:, :, c:], ones, heads, d, t, t, 1 / math.sqrt(d), **kw).float()
    assert float((o - 1).abs().max()) < 1e-2
    vt = _bf(torch.randn(b, c, t, generator=g))
    o1 = ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, 1 / math.sqrt(d), **kw).float()
    perm = torch.randperm(t, generator=g).cuda()
    qkv2 = qkv.clone()
    qkv2[:, :, c:2 * c] = qkv[:, perm, c:2 * c]                                           # keys permuted, queries untouched
    o2 = ops.attention(qkv2, qkv2[:, :, c:], vt[:, :, perm].contiguous(), heads, d, t, t, 1 / math.sqrt(d), **kw).float()
    assert rel_l2(o2.cpu(), o1.cpu()) < 6e-3


@pytest.mark.parametrize("b,hw,c", [(8, 64, 320), (8, 512, 128), (8, 8, 1280)])
def test_groupnorm_groups_are_standardised(ops, b, hw, c):
    g = torch.Generator().manual_seed(c + hw)
    x = _bf(torch.randn(b, hw, hw, c, generator=g) * 3 + 1.5)
    y = ops.group_norm(x, torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), 32, 1e-5, False).float()
    yg = y.view(b, hw * hw, 32, c // 32).permute(0, 2, 1, 3).reshape(b, 32, -1)
    assert float(yg.mean(-1).abs().max()) < 2e-2 and float((yg.var(-1, unbiased=False) - 1).abs().max()) < 2e-2


def test_full_size_model_is_image_independent():
    """Full-size architecture (random weights), 2 DDIM steps: a batch of two equals the two images run alone."""
    import bench
    dev = torch.device("cuda", 0)
    m = bench.build_model(2, dev, 0, 1)
    g = torch.Generator(device=dev).manual_seed(3)
    img = torch.rand(2, 3, 512, 512, generator=g, device=dev)
    nz = (torch.randn(2, 4, 64, 64, generator=g, device=dev), torch.randn(2, 4, 64, 64, generator=g, device=dev))
    both = m(img, "ir", noise=nz).clone()
    for i in range(2):
        one = m(img[i:i + 1], "ir", noise=(nz[0][i:i + 1], nz[1][i:i + 1]))
        assert rel_l2(one.cpu(), both[i:i + 1].cpu()) < 1.5e-2          # different tile shapes / reduction orders per batch size
    assert bool(torch.isfinite(both).all())
