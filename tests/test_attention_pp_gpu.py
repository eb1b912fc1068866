"""Ping-pong attention kernel (csrc/attention_pp.hip, the d = 64 self-attention path of ur_attention_fwd) against an fp64 torch
softmax(QK^T / sqrt(d)) V - reference call sites: diffusers Attention via /root/reference/src/modules/diffuie/base_model.py:138,159,191
and controller.py:101-141.  Covers both 16-bit types, the production layouts (q | k interleaved in one QKV tensor, V transposed),
every tile-count class of the 4-tile unrolled loop, the slow path of the online softmax (reference jumps at chosen tiles, all-negative
rows, rows that never trigger it), run-to-run bit identity, and agreement with the round-1 kernel (UR_ATTN_NOPP=1, second process)."""
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    from unirestore_amd import ops as o
    return o


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _ref(q, k, v, heads, d, scale):
    b, t, _ = q.shape
    qh, kh, vh = (x.double().view(b, -1, heads, d).transpose(1, 2) for x in (q, k, v))
    return (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(1, 2).reshape(b, t, heads * d)


def _run(ops, q, k, v, heads, d, scale, dt, packed=True):
    """packed: q | k | (unused) in one [B, T, 3C] tensor as the fused QKV GEMM writes them (ldq = ldk = 3C)"""
    b, t, c = q.shape
    vt = v.transpose(1, 2).contiguous().to(dt).cuda()
    if packed:
        qkv = torch.cat([q, k, torch.zeros_like(q)], -1).to(dt).cuda()
        return ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, scale, ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c,
                             bs_vt=c * t, batch=b)
    return ops.attention(q.to(dt).cuda(), k.to(dt).cuda(), vt, heads, d, t, t, scale, ldq=c, ldk=c, bs_q=t * c, bs_k=t * c,
                         bs_vt=c * t, batch=b)


TOL = {torch.bfloat16: 6e-3, torch.float16: 8e-4}     # the attention tolerances of tests/test_ops_gpu.py (P is a 16-bit operand)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("b,heads,t", [(1, 1, 256), (2, 5, 512), (1, 2, 768), (2, 5, 1024), (8, 5, 4096)])
def test_pp_attention_matches_fp64(ops, dt, b, heads, t):
    d = 64
    g = torch.Generator().manual_seed(b * 1000 + t)
    q, k, v = (torch.randn(b, t, heads * d, generator=g).to(dt).float() for _ in range(3))
    ref = _ref(q, k, v, heads, d, 1 / math.sqrt(d))
    o = _run(ops, q, k, v, heads, d, 1 / math.sqrt(d), dt, packed=(t != 768))
    assert rel_l2(o.cpu(), ref) < TOL[dt]
    again = _run(ops, q, k, v, heads, d, 1 / math.sqrt(d), dt, packed=(t != 768))
    assert torch.equal(o, again)                                     # no atomics, fixed order: bit-identical


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_pp_attention_online_softmax_slow_path(ops, dt):
    """Reference jumps: a key in tile 3, a larger one in tile 9 and a huge one in the last tile each raise some rows' maximum
    by far more than the 2^14 head-room of the fast path; other rows never leave it; one row has only very negative scores."""
    d, t, heads = 64, 1024, 2
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(1, t, heads * d, generator=g) for _ in range(3))
    k[0, 3 * 64 + 7, :d] = q[0, 5, :d] * 6          # row 5, head 0: score ~ 6 |q|^2 / 8 ~ 48
    k[0, 9 * 64 + 1, :d] = q[0, 5, :d] * 12         # again, higher, later
    k[0, t - 1, d:] = q[0, 300, d:] * 30            # head 1, row 300: ~ 240 in the last tile
    q[0, 700, :d] *= 40                              # a row with a wide score range from tile 0 on
    # production convention (modules/nn.py Q_FOLD): the softmax scale * log2(e) is folded into q BEFORE its rounding to 16 bits and
    # the kernel is called with scale = ln 2.  (The in-kernel fallback for other scales rounds q a second time: fine for ordinary
    # scores - the other tests - but on scores of a few hundred the second bf16 rounding alone moves single weights by ~5 %.)
    q = q * (math.log2(math.e) / math.sqrt(d))
    q, k, v = (x.to(dt).float() for x in (q, k, v))
    ref = _ref(q, k, v, heads, d, math.log(2.0))
    o = _run(ops, q, k, v, heads, d, math.log(2.0), dt)
    assert bool(torch.isfinite(o).all())
    assert rel_l2(o.cpu(), ref) < TOL[dt]
    # every row separately: a wrong rescale corrupts single rows, which a global norm would hide
    err = (o.cpu().double() - ref).view(t, heads, d).norm(dim=-1) / ref.view(t, heads, d).norm(dim=-1).clamp_min(1e-3)
    assert float(err.max()) < 10 * TOL[dt]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_pp_attention_all_scores_far_below_zero(ops, dt):
    """q.k = -|q|^2 * 20 for every key: the first tile must set a NEGATIVE reference (exp2 underflow would give 0 / 0)."""
    d, t = 64, 256
    g = torch.Generator().manual_seed(6)
    q = torch.randn(1, t, d, generator=g).to(dt).float()
    base = torch.randn(1, 1, d, generator=g)
    q = (base + 0.01 * q).to(dt).float()
    k = (-20 * base + 0.01 * torch.randn(1, t, d, generator=g)).to(dt).float()
    v = torch.randn(1, t, d, generator=g).to(dt).float()
    ref = _ref(q, k, v, 1, d, 1.0)
    o = _run(ops, q, k, v, 1, d, 1.0, dt, packed=False)
    assert bool(torch.isfinite(o).all())
    assert rel_l2(o.cpu(), ref) < 4 * TOL[dt]         # scores ~ -1300 +- 3: the 16-bit q / k rounding is amplified by the scale


def test_pp_attention_scale_folded_equals_scale_passed(ops):
    """scale * log2(e) == 1 skips the in-kernel Q pre-multiplication: q pre-scaled by the caller gives the same attention."""
    d, t, heads, dt = 64, 512, 2, torch.float16
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(2, t, heads * d, generator=g).to(dt).float() for _ in range(3))
    ref = _ref(q, k, v, heads, d, 0.125)
    c = 0.125 * math.log2(math.e)
    o = _run(ops, q * c, k, v, heads, d, math.log(2.0), dt)       # scale * log2(e) = 1
    assert rel_l2(o.cpu(), ref) < 2 * TOL[dt]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_pp_attention_key_split_last_round(ops, dt):
    """B*H*(T/256) = 640 workgroups = 2.5 rounds of the 256 CUs: with a workspace the last 128 query tiles run as 256 key halves +
    the merge kernel (ur_attention_fwd_ws); without one (ur_attention_fwd) unsplit.  Both must match fp64, and each other closely."""
    from unirestore_amd.capi import lib
    b, heads, t, d = 32, 5, 1024, 64
    c = heads * d
    assert lib.ur_attention_workspace_bytes(b, heads, t, t, d) == 128 * 2 * 256 * 68 * 4
    assert lib.ur_attention_workspace_bytes(b, 4, t, t, d) == 0                     # 512 workgroups: whole rounds
    g = torch.Generator().manual_seed(21)
    q, k, v = (torch.randn(b, t, c, generator=g).to(dt).float() for _ in range(3))
    k[3, 700, :d] = q[3, 40, :d] * 9                                                 # a reference jump inside the SECOND key half
    k = k.to(dt).float()
    ref = _ref(q, k, v, heads, d, 0.125)
    o_split = _run(ops, q, k, v, heads, d, 0.125, dt)                                # ops.attention passes the workspace
    qkv = torch.cat([q, k, torch.zeros_like(q)], -1).to(dt).cuda()
    vt = v.transpose(1, 2).contiguous().to(dt).cuda()
    o_plain = torch.empty_like(o_split)
    rc = lib.ur_attention_fwd(qkv.data_ptr(), qkv[:, :, c:].data_ptr(), vt.data_ptr(), o_plain.data_ptr(), b, heads, t, t, d, 3 * c, 3 * c, t,
                              c, t * 3 * c, t * 3 * c, c * t, t * c, 0.125, ops._dt(qkv), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_l2(o_split.cpu(), ref) < TOL[dt] and rel_l2(o_plain.cpu(), ref) < TOL[dt]
    assert rel_l2(o_split.cpu(), o_plain.cpu()) < TOL[dt]
    head_rows = o_split[:, :, :].view(b * t, c)
    assert not torch.equal(o_split, o_plain)                                         # the split path really ran
    assert torch.equal(o_split[:25], o_plain[:25])                                   # ... and only on the trailing tiles (tile 512 = image 25, head 3)
    assert torch.equal(o_split, _run(ops, q, k, v, heads, d, 0.125, dt))             # bit-reproducible


_CHILD = r"""
import sys, math, torch
sys.path.insert(0, {root!r})
from unirestore_amd import ops
g = torch.Generator().manual_seed(11)
b, t, heads, d = 2, 1024, 5, 64
c = heads * d
qkv = torch.randn(b, t, 3 * c, generator=g).to(torch.bfloat16).cuda()
vt = torch.randn(b, c, t, generator=g).to(torch.bfloat16).cuda()
o = ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, 0.125, ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=b)
torch.save(o.float().cpu(), sys.argv[1])
"""


def test_pp_attention_agrees_with_round1_kernel(tmp_path):
    outs = []
    for nopp in ("0", "1"):
        f = tmp_path / f"o{nopp}.pt"
        env = dict(os.environ, UR_ATTN_NOPP=nopp)
        subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), str(f)], check=True, env=env, timeout=300)
        outs.append(torch.load(f))
    assert not torch.equal(outs[0], outs[1])                 # two different kernels really ran
    assert rel_l2(outs[0], outs[1]) < 6e-3
