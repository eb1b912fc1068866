"""Leaf parameter holders + the SD building blocks, executed through the HIP ops (NHWC bf16).

The classes keep the reference's / HF checkpoints' parameter names so `load_state_dict`
of the published weights works unchanged; fp32 masters stay on the host, `packed()` builds the bf16
device copies the kernels read.  Calling a leaf's torch `forward` is an error: there is no eager path.

Block semantics restated from SURVEY.md Appendix C (diffusers 0.29.0 is not vendored in the reference):
reference call sites are base_model.py:94-209, controller.py:101-170, autoencoder.py:11-72.
"""
import math
from typing import Optional

import os

import torch
import torch.nn as nn

from .. import chain, ops
from ..ops import UR_ACT_GEGLU, UR_ACT_GELU, UR_ACT_NONE, UR_ACT_SILU

DEV = "cuda"
FUSE_LN = __import__("os").environ.get("UR_FUSE_LN", "1") == "1"   # LayerNorm folded into the consuming GEMMs
# Token-stationary fused chains (csrc/tchain.hip) for the 320-channel transformer blocks: 3 launches per Transformer2DModel
# (HEAD chain, self-attention, TAIL chain) instead of 12.  UR_CHAIN=0 selects the per-layer path (A/B, and the shapes the
# chains do not cover take it anyway).
CHAIN = os.environ.get("UR_CHAIN", "1") == "1"


class _NoEager:
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__}: no eager/PyTorch path exists; use the HIP-backed graph")


class Conv2d(_NoEager, nn.Conv2d):
    def packed(self, pair=False, scale: Optional[torch.Tensor] = None) -> ops.PackedConv:
        key = ("pk", pair, ops.act_dtype())           # device copies are per 16-bit type
        if key not in self.__dict__:
            w, b = self.weight, self.bias
            if scale is not None:                     # fold a per-output-channel scale (NAFBlock beta/gamma)
                s = scale.detach().reshape(-1)
                w = w * s[:, None, None, None]
                b = None if b is None else b * s
            self.__dict__[key] = ops.pack_conv(w, b, DEV, pair=pair, groups=self.groups)
        return self.__dict__[key]


class Linear(_NoEager, nn.Linear):
    def packed(self, pair=False) -> ops.PackedConv:
        key = ("pk", pair, ops.act_dtype())
        if key not in self.__dict__:
            self.__dict__[key] = ops.pack_conv(self.weight, self.bias, DEV, pair=pair)
        return self.__dict__[key]

    def dev_f32(self):
        if "f32" not in self.__dict__:
            self.__dict__["f32"] = (self.weight.detach().float().to(DEV).contiguous(),
                                    None if self.bias is None else self.bias.detach().float().to(DEV).contiguous())
        return self.__dict__["f32"]


class _Affine:
    def dev(self):
        if "aff" not in self.__dict__:
            self.__dict__["aff"] = (self.weight.detach().float().to(DEV).contiguous(),
                                    self.bias.detach().float().to(DEV).contiguous())
        return self.__dict__["aff"]


class GroupNorm(_NoEager, _Affine, nn.GroupNorm):
    def run(self, x, silu=False, x2=None):
        g, b = self.dev()
        return ops.group_norm(x, g, b, self.num_groups, self.eps, silu, x2=x2)

    def coeffs(self, x, x2=None):
        """fp32 ab [N][2][C]: GroupNorm(x | x2) == a*x + b per (image, channel) - for a consumer that applies it itself."""
        g, b = self.dev()
        return ops.gn_finalize(x, g, b, self.num_groups, self.eps, x2=x2)


class LayerNorm(_NoEager, _Affine, nn.LayerNorm):
    def run(self, x):
        g, b = self.dev()
        return ops.layer_norm(x, g, b, self.eps)


def invalidate_packed(model: nn.Module):
    """Drop every cached device copy (call after loading new weights)."""
    for m in model.modules():
        for k in [k for k in m.__dict__ if k in ("aff", "f32", "dw", "vecs") or
                  (isinstance(k, tuple) and k[0] in ("pk", "cache", "fused", "ctx", "gb"))]:
            del m.__dict__[k]


def sinusoid_table(timesteps, dim: int) -> torch.Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) for a list of integer timesteps -> fp32 [S, dim] = [cos|sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = torch.as_tensor(list(timesteps), dtype=torch.float32).reshape(-1, 1) * freqs.reshape(1, -1)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, time_dim):
        super().__init__()
        self.linear_1 = Linear(in_dim, time_dim)
        self.linear_2 = Linear(time_dim, time_dim)

    def silu_emb(self, sin_table: torch.Tensor) -> torch.Tensor:
        """silu(linear_2(silu(linear_1(sin)))) for all rows: the only form the resnets consume."""
        w1, b1 = self.linear_1.dev_f32()
        w2, b2 = self.linear_2.dev_f32()
        h = ops.linear_f32(sin_table.to(DEV), w1, b1, UR_ACT_SILU)
        return ops.linear_f32(h, w2, b2, UR_ACT_SILU)


# ----------------------------------------------------------------------------------------------- resnet
class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels=None, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = GroupNorm(groups, cin, eps=eps)
        self.conv1 = Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = Linear(temb_channels, cout) if temb_channels else None
        self.norm2 = GroupNorm(groups, cout, eps=eps)
        self.conv2 = Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = Conv2d(cin, cout, 1) if cin != cout else None
        self.tbias = None           # fp32 [S or B, cout]: conv1.bias + time_emb_proj(silu(emb)) rows

    def set_time_table(self, silu_emb: torch.Tensor):
        """Per-row conv1 bias with the time embedding folded in (rows = schedule steps, or samples)."""
        for k in [k for k in self.__dict__ if isinstance(k, tuple) and k[:2] == ("cache", "tb_all")]:
            del self.__dict__[k]
        w, b = self.time_emb_proj.dev_f32()
        cb = self.conv1.bias.detach().float().to(DEV)
        self.tbias = ops.linear_f32(silu_emb, w, b + cb)

    def _tbias_all(self, b):
        """[S*b, cout] bias rows for a schedule-batched pass (image n = step*b + i uses the row of its step)."""
        key = ("cache", "tb_all", b)
        if key not in self.__dict__:
            self.__dict__[key] = self.tbias.repeat_interleave(b, dim=0).contiguous()
        return self.__dict__[key]

    def run(self, x, x2=None, step=None, sample_bias=None, control=None):
        """x (+x2: virtual concat) NHWC bf16.  step: row of the time table, or "all" when the batch stacks every step of
        the schedule (step-major); sample_bias: explicit [N,cout] per-image rows; control: {width: NHWC map} for a grafted
        SPADE (base_model.py:56-92)."""
        bias = None
        if self.time_emb_proj is not None:
            if sample_bias is not None:
                bias = sample_bias
            elif isinstance(step, str):
                bias = self._tbias_all(x.shape[0] // self.tbias.shape[0])
            else:
                bias = self.tbias[step]
        h = gn_silu_conv(self.norm1, x, self.conv1.packed(), x2=x2, bias=bias, gn=True)
        pc2 = self.conv2.packed()
        if self.conv_shortcut is not None:    # (tried as a parallel graph branch at the low-resolution levels: fork/join cost more than it hid)
            sc = ops.conv(x, self.conv_shortcut.packed(), x2=x2)
        else:
            sc = x
        if control is not None and "spade" in self._modules:
            if x2 is not None and self.conv_shortcut is None:
                raise ValueError("identity shortcut cannot take a virtual concat")
            h = gn_silu_conv(self.norm2, h, pc2, gn=True)
            return self.spade.run(h, control[h.shape[2]], residual=sc)          # (no fused sums: the next norm takes its own)
        return gn_silu_conv(self.norm2, h, pc2, residual=sc, gn=True)


FUSE_GN = os.environ.get("UR_FUSE_GN", "1") == "1"       # GroupNorm apply + SiLU inside the consuming 3x3 conv's loader
# Measured (tools/ab_micro.py, MI355X): the in-LDS pass costs the halo conv ~15 us per launch on top of its MFMA time (the two
# waves of a SIMD meet at a barrier every tap, so little of the VALU work hides), which beats the separate apply pass (HBM read
# + write of the whole activation) only on the VAE's large maps: 128 ch @ 512x512 245 vs 253 us, 320 ch @ 64x64 85 vs 79.5 us.
# The in-loader pass is repeated by every output-channel tile of the same pixels (and by neighbouring patches for the halo), so
# it only pays where the conv has one or two channel tiles: Cout <= 256 (VAE levels, Controller), not the UNet's 320-1280.
FUSE_GN_MIN_PIXELS = int(os.environ.get("UR_FUSE_GN_MIN_PIXELS", str(256 * 256)))
FUSE_GN_MAX_COUT = int(os.environ.get("UR_FUSE_GN_MAX_COUT", "256"))


def gn_silu_conv(norm: "GroupNorm", x, pc, x2=None, **kw):
    """conv(SiLU(GroupNorm(x | x2))): statistics -> per-(image, channel) affine, then either the conv applies it while it
    loads its input patch (no normalised tensor in HBM) or a separate apply pass runs first."""
    ab = norm.coeffs(x, x2=x2)
    if FUSE_GN and x.shape[1] * x.shape[2] >= FUSE_GN_MIN_PIXELS and pc.cout_out <= FUSE_GN_MAX_COUT and \
            ops.conv_plan(x, pc, x2=x2, gn=kw.get("gn", False), gn_ab=True, residual=kw.get("residual") is not None).prologue_ok:
        return ops.conv(x, pc, x2=x2, gn_ab=ab, gn_silu=True, **kw)
    return ops.conv(ops.gn_apply(x, ab, silu=True, x2=x2), pc, **kw)


class Downsample2D(nn.Module):
    def __init__(self, c, padding):
        super().__init__()
        self.padding = padding
        self.conv = Conv2d(c, c, 3, stride=2, padding=padding)

    def run(self, x):
        n, h, w, _ = x.shape
        if self.padding == 0:          # VAE encoder: zero pad right/bottom by one, then stride-2 VALID conv
            return ops.conv(x, self.conv.packed(), stride=2, pad=(0, 0), out_hw=(h // 2, w // 2), gn=True)
        return ops.conv(x, self.conv.packed(), stride=2, pad=(1, 1), gn=True)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3, padding=1)

    def run(self, x):
        return ops.conv(x, self.conv.packed(), upsample=True, gn=True)      # nearest-2x gather fused into the loader


# ----------------------------------------------------------------------------------------------- attention
class _ToOut(nn.ModuleList):
    def __init__(self, c):
        super().__init__([Linear(c, c), nn.Identity()])


# Softmax scale folded into the query projection.  The d = 64 attention kernels work in the exp2 domain on scores that are already
# multiplied by scale * log2(e) (csrc/attention_pp.hip: the subtraction of the running maximum is the MFMA's C operand, so a
# per-score multiply would be the only arithmetic left in front of the exponential).  Folding the factor into the fp32 master of
# to_q BEFORE its one rounding to 16 bits costs no accuracy; the kernel's own fallback - multiply the 16-bit q by the factor and
# round again - does (bf16: a second 2^-9 rounding of every q element).  ur_attention_fwd is then called with scale = ln 2, i.e.
# scale * log2(e) = 1.
Q_FOLD = math.log2(math.e)
LN2 = math.log(2.0)


def _scaled_first(ts, q_scale):
    ts = [None if t is None else t.detach().float() for t in ts]
    if q_scale != 1.0 and ts[0] is not None:
        ts[0] = ts[0] * q_scale
    return ts


def _fused_qkv(mod, names, q_scale=1.0):
    ck = ("cache", "qkv", ops.act_dtype(), q_scale)
    if ck not in mod.__dict__:
        ws = _scaled_first([getattr(mod, n).weight for n in names], q_scale)
        bs = _scaled_first([getattr(mod, n).bias for n in names], q_scale)
        mod.__dict__[ck] = ops.pack_conv(torch.cat(ws, 0), None if bs[0] is None else torch.cat(bs, 0), DEV)
    return mod.__dict__[ck]


def _fused_ln(mod, key, names, norm, pair=False, q_scale=1.0):
    """Linear(LayerNorm(x)) weights folded for the LN-fused GEMM epilogue (cached): rows = cat of `names`."""
    ck = ("cache", "ln", key, ops.act_dtype(), q_scale)
    if ck not in mod.__dict__:
        ws = _scaled_first([getattr(mod, n).weight for n in names], q_scale)
        bs = _scaled_first([getattr(mod, n).bias for n in names], q_scale)
        mod.__dict__[ck] = ops.pack_linear_ln(torch.cat(ws, 0), None if bs[0] is None else torch.cat(bs, 0), norm.weight, norm.bias, norm.eps,
                                               DEV, pair=pair)
    return mod.__dict__[ck]


NO_FLASH512 = os.environ.get("UR_NO_FLASH512", "0") == "1"     # A/B switch: chunked GEMM form for the d = 512 VAE attention


def self_attention(mod, h, heads, residual, gn_kw={}, ln=None, rows=False):
    """h: [B,T,C] bf16.  One fused QKV GEMM (V written transposed), flash attention, output projection with the
    residual in its epilogue.  ln=(norm, row_stats): h is the RAW residual stream and LayerNorm is folded into the QKV
    GEMM; otherwise h is already normalised.  rows=True makes the output projection emit per-row sums for the next LN."""
    b, t, c = h.shape
    d = c // heads
    ldvt = ops.round_up(t, 8)
    vt = torch.zeros((b, c, ldvt), dtype=h.dtype, device=h.device) if ldvt != t else \
        torch.empty((b, c, ldvt), dtype=h.dtype, device=h.device)
    scale = 1.0 / math.sqrt(d)
    qs = scale * Q_FOLD if d == 64 else 1.0           # d = 64: to_q carries scale * log2(e) (see Q_FOLD above)
    if ln is not None:
        qk = ops.linear(h, _fused_ln(mod, "qkv", ("to_q", "to_k", "to_v"), ln[0], q_scale=qs), ln_stats=ln[1], yt=vt, n_split=2 * c, t_rows=t)
    else:
        qk = ops.linear(h, _fused_qkv(mod, ("to_q", "to_k", "to_v"), q_scale=qs), yt=vt, n_split=2 * c, t_rows=t)   # [B,T,3C] (V cols unused)
    if d in (64, 128) or (d == 512 and not NO_FLASH512):
        o = ops.attention(qk, qk[:, :, c:], vt, heads, d, t, t, LN2 if d == 64 else scale, ldq=3 * c, ldk=3 * c,
                          bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * ldvt, batch=b)
    else:
        o = attention_gemm(qk[:, :, :c], qk[:, :, c:2 * c], vt, heads, d, t)
    return ops.linear(o, mod.to_out[0].packed(), residual=residual, rows=rows, **gn_kw)


def attention_gemm(q, k, vt, heads, d, t, s_bytes=256 << 20):
    """Large-head-dim attention (VAE mid block: 1 head x 512) as S = QK^T (fp32) -> row softmax -> P V, in QUERY CHUNKS so
    that the fp32 score block stays bounded (<= s_bytes = 256 MB, instead of B x T x T x 4 = 1 GiB per image at 1024x1024)."""
    b = q.shape[0]
    rows = max(256, min(t, (s_bytes // (4 * b * t)) // 256 * 256))
    out = torch.empty((b, t, heads * d), dtype=q.dtype, device=q.device)
    for hh in range(heads):
        qh, kh, vh = q[:, :, hh * d:(hh + 1) * d], k[:, :, hh * d:(hh + 1) * d], vt[:, hh * d:(hh + 1) * d, :]
        for r0 in range(0, t, rows):
            r1 = min(t, r0 + rows)
            s = ops.bmm_nt(qh[:, r0:r1], kh, out_f32=True, out_scale=1.0 / math.sqrt(d))
            p = ops.softmax_rows(s, ldp=vt.shape[-1])
            out[:, r0:r1, hh * d:(hh + 1) * d] = ops.bmm_nt(p, vh)
    return out


class AttentionBlock(nn.Module):
    """Legacy spatial self-attention (VAE mid block; Controller AttnDownBlock2D / UNetMidBlock2D)."""

    def __init__(self, c, head_dim, groups=32, eps=1e-5):
        super().__init__()
        self.heads = c // head_dim
        self.group_norm = GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = Linear(c, c), Linear(c, c), Linear(c, c)
        self.to_out = _ToOut(c)

    def run(self, x):
        n, hh, ww, c = x.shape
        h = self.group_norm.run(x).view(n, hh * ww, c)
        o = self_attention(self, h, self.heads, x.view(n, hh * ww, c), gn_kw=dict(gn=True, gn_hw=(n, hh * ww)))
        return ops.carry(o, o.view(n, hh, ww, c))


class CrossAttention(nn.Module):
    def __init__(self, c, heads, kv_dim=None):
        super().__init__()
        self.heads = heads
        kv_dim = kv_dim or c
        self.to_q = Linear(c, c, bias=False)
        self.to_k = Linear(kv_dim, c, bias=False)
        self.to_v = Linear(kv_dim, c, bias=False)
        self.to_out = _ToOut(c)

    def context_kv(self, ctx: torch.Tensor):
        """K [Tk,C] and V^T [C,ldvt] of the (constant) context, computed once (base_model.py:23-27,221)."""
        key = ("cache", "ctx", ops.act_dtype())
        if key not in self.__dict__:
            tk = ctx.shape[1]
            c = self.to_q.out_features
            kv = _fused_qkv(self, ("to_k", "to_v"))
            vt = torch.zeros((1, c, ops.round_up(tk, 8)), dtype=ctx.dtype, device=DEV)
            k = ops.linear(ctx, kv, yt=vt, n_split=c, t_rows=tk)      # [1,Tk,2C]; K = first C columns
            self.__dict__[key] = (k, vt, tk)
        return self.__dict__[key]

    def run_cross(self, h, ctx, residual, ln=None, rows=False):
        b, t, c = h.shape
        d = c // self.heads
        k, vt, tk = self.context_kv(ctx)
        if ln is not None:
            q = ops.linear(h, _fused_ln(self, "q", ("to_q",), ln[0]), ln_stats=ln[1])
        else:
            q = ops.linear(h, self.to_q.packed())
        o = ops.attention(q, k, vt, self.heads, d, t, tk, 1.0 / math.sqrt(d), ldq=c, ldk=2 * c, bs_q=t * c, bs_k=0,
                          bs_vt=0, batch=b)
        return ops.linear(o, self.to_out[0].packed(), residual=residual, rows=rows)


class GEGLU(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.proj = Linear(c, inner * 2)


class FeedForward(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(c, 4 * c), nn.Identity(), Linear(4 * c, c)])

    def run(self, h, residual, ln=None):
        if ln is not None:
            a = ops.linear(h, _fused_ln(self.net[0], "proj", ("proj",), ln[0], pair=True), ln_stats=ln[1], act=UR_ACT_GEGLU)
        else:
            a = ops.linear(h, self.net[0].proj.packed(pair=True), act=UR_ACT_GEGLU)     # a * gelu(g) in the epilogue
        return ops.linear(a, self.net[2].packed(), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, c, heads, cross_dim):
        super().__init__()
        self.norm1, self.attn1 = LayerNorm(c), CrossAttention(c, heads)
        self.norm2, self.attn2 = LayerNorm(c), CrossAttention(c, heads, cross_dim)
        self.norm3, self.ff = LayerNorm(c), FeedForward(c)

    def run(self, h, ctx):
        st = ops.ln_of(h)
        if st is None:                                  # no producer-side row sums: plain LayerNorm passes
            h = self_attention(self.attn1, self.norm1.run(h), self.attn1.heads, h)
            h = self.attn2.run_cross(self.norm2.run(h), ctx, h)
            return self.ff.run(self.norm3.run(h), h)
        # LayerNorm folded into the consuming GEMMs; each residual-producing GEMM leaves the next LN's row sums
        h = self_attention(self.attn1, h, self.attn1.heads, h, ln=(self.norm1, st), rows=True)
        h = self.attn2.run_cross(h, ctx, h, ln=(self.norm2, ops.ln_of(h)), rows=True)
        return self.ff.run(h, h, ln=(self.norm3, ops.ln_of(h)))


class Transformer2DModel(nn.Module):
    def __init__(self, c, heads, cross_dim, groups=32):
        super().__init__()
        self.norm = GroupNorm(groups, c, eps=1e-6)
        self.proj_in = Linear(c, c)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(c, heads, cross_dim)])
        self.proj_out = Linear(c, c)

    def _chain_streams(self, ctx):
        """Weight streams of the HEAD / TAIL chains for the current 16-bit type (packed once; dropped by invalidate_packed)."""
        key = ("cache", "chain", ops.act_dtype())
        if key not in self.__dict__:
            b = self.transformer_blocks[0]
            ff1, ff2 = b.ff.net[0].proj, b.ff.net[2]
            qs = Q_FOLD / math.sqrt(self.proj_in.weight.shape[0] // b.attn1.heads)          # softmax scale * log2(e) folded into to_q
            head = chain.pack_head(self.proj_in.weight, self.proj_in.bias, b.attn1.to_q.weight.detach().float() * qs, b.attn1.to_k.weight,
                                   b.attn1.to_v.weight, b.norm1.weight, b.norm1.bias, DEV)
            tail = chain.pack_tail(b.attn1.to_out[0].weight, b.attn1.to_out[0].bias, b.attn2.to_q.weight, b.norm2.weight, b.norm2.bias,
                                   b.attn2.to_k.weight, b.attn2.to_v.weight, ctx[0].float(), b.attn2.to_out[0].weight, b.attn2.to_out[0].bias,
                                   ff1.weight, ff1.bias, ff2.weight, ff2.bias, b.norm3.weight, b.norm3.bias,
                                   self.proj_out.weight, self.proj_out.bias, b.attn2.heads, DEV)
            self.__dict__[key] = (head, tail)
        return self.__dict__[key]

    def _chain_ok(self, x, ctx):
        b = self.transformer_blocks[0]
        n, hh, ww, c = x.shape
        return (CHAIN and c == chain.CHAIN_C and (hh * ww) % chain.CHAIN_TOK == 0 and b.attn1.heads == 5 and b.attn2.heads == 5 and
                ctx.shape[0] == 1 and ctx.shape[1] <= 80 and b.attn1.to_q.bias is None and b.attn2.to_q.bias is None and
                b.norm1.eps == b.norm2.eps == b.norm3.eps and ff_hidden(b) == 4 * c and x.is_contiguous())

    def run(self, x, ctx):
        n, hh, ww, c = x.shape
        if self._chain_ok(x, ctx):
            # HEAD chain: GroupNorm apply + proj_in + LayerNorm1-folded q / k / v^T; flash self-attention; TAIL chain: everything else,
            # leaving the GroupNorm partial sums of the next resnet's norm1 (base_model.py:137-160,184-198 via diffusers)
            b = self.transformer_blocks[0]
            head, tail = self._chain_streams(ctx)
            t, heads = hh * ww, b.attn1.heads
            d = c // heads
            h0, q, k, vt = chain.transformer_head_fused(x, self.norm.coeffs(x), head, n, b.norm1.eps)
            o1 = ops.attention(q, k, vt, heads, d, t, t, LN2, ldq=c, ldk=c, bs_q=t * c, bs_k=t * c, bs_vt=c * t, batch=n)   # (q pre-scaled)
            y = chain.transformer_tail_fused(o1, h0, x, tail, n, 4 * c, heads, ctx.shape[1], b.norm1.eps, 1.0 / math.sqrt(d))
            return ops.carry(y, y.view(n, hh, ww, c))
        h = ops.linear(self.norm.run(x).view(n, hh * ww, c), self.proj_in.packed(), rows=FUSE_LN)
        h = self.transformer_blocks[0].run(h, ctx)
        o = ops.linear(h, self.proj_out.packed(), residual=x.view(n, hh * ww, c), gn=True, gn_hw=(n, hh * ww))
        return ops.carry(o, o.view(n, hh, ww, c))


def ff_hidden(block) -> int:
    return block.ff.net[2].in_features


# ----------------------------------------------------------------------------------------------- blocks
class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, attn=None, heads=None, head_dim=None, cross_dim=None,
                 add_downsample=True, layers=2, groups=32, eps=1e-5):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])
        self.attn_kind = attn
        if attn == "cross":
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, groups) for _ in range(layers)])
        elif attn == "self":
            self.attentions = nn.ModuleList([AttentionBlock(cout, head_dim, groups, eps) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, 1)]) if add_downsample else None


class MidBlock(nn.Module):
    def __init__(self, c, temb, attn, heads=None, head_dim=None, cross_dim=None, groups=32, eps=1e-5):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attn_kind = attn
        if attn == "cross":
            self.attentions = nn.ModuleList([Transformer2DModel(c, heads, cross_dim, groups)])
        else:
            self.attentions = nn.ModuleList([AttentionBlock(c, head_dim, groups, eps)])

    def run(self, h, step=None, ctx=None, sample_bias=None, control=None):
        sb = sample_bias or (None, None)
        h = self.resnets[0].run(h, step=step, sample_bias=sb[0], control=control)
        h = self.attentions[0].run(h, ctx) if self.attn_kind == "cross" else self.attentions[0].run(h)
        return self.resnets[1].run(h, step=step, sample_bias=sb[1], control=control)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb, attn=None, heads=None, cross_dim=None,
                 add_upsample=True, layers=3, groups=32, eps=1e-5):
        super().__init__()
        res = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(res)
        self.attn_kind = attn
        if attn == "cross":
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, groups) for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None


class UNet2DConditionModel(nn.Module):
    """SD-2.1 UNet parameter tree (HF names).  Walked by ControlledUNet, as base_model.py:94-209 does."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 heads=(5, 10, 20, 20), cross_dim=1024, groups=32, layers=2):
        super().__init__()
        ch = list(block_out_channels)
        self.time_proj_dim = ch[0]
        temb = ch[0] * 4
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.conv_in = Conv2d(in_channels, ch[0], 3, padding=1)
        nb = len(ch)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i in range(nb):
            cin, out = out, ch[i]
            last = i == nb - 1
            self.down_blocks.append(DownBlock(cin, out, temb, attn=None if last else "cross", heads=heads[i],
                                              cross_dim=cross_dim, add_downsample=not last, layers=layers, groups=groups))
        self.mid_block = MidBlock(ch[-1], temb, "cross", heads=heads[-1], cross_dim=cross_dim, groups=groups)
        rev, rheads = ch[::-1], list(heads)[::-1]
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i in range(nb):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, nb - 1)]
            self.up_blocks.append(UpBlock(cin, out, prev, temb, attn=None if i == 0 else "cross", heads=rheads[i],
                                          cross_dim=cross_dim, add_upsample=i < nb - 1, layers=layers + 1, groups=groups))
        self.conv_norm_out = GroupNorm(groups, ch[0], eps=1e-5)
        self.conv_out = Conv2d(ch[0], out_channels, 3, padding=1)


# ----------------------------------------------------------------------------------------------- VAE
class EncDownBlock(nn.Module):
    def __init__(self, cin, cout, add_downsample, groups, layers=2):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, 0)]) if add_downsample else None

    def run(self, h):
        for r in self.resnets:
            h = r.run(h)
        return self.downsamplers[0].run(h) if self.downsamplers is not None else h


class DecUpBlock(nn.Module):
    def __init__(self, cin, cout, add_upsample, groups, layers=3):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def run(self, h):
        for r in self.resnets:
            h = r.run(h)
        return self.upsamplers[0].run(h) if self.upsamplers is not None else h


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, ch, groups):
        super().__init__()
        self.conv_in = Conv2d(in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i in range(len(ch)):
            cin, out = out, ch[i]
            self.down_blocks.append(EncDownBlock(cin, out, i < len(ch) - 1, groups))
        self.mid_block = MidBlock(ch[-1], None, "self", head_dim=ch[-1], groups=groups, eps=1e-6)
        self.conv_norm_out = GroupNorm(groups, ch[-1], eps=1e-6)
        self.conv_out = Conv2d(ch[-1], 2 * latent_channels, 3, padding=1)


class Decoder(nn.Module):
    def __init__(self, out_channels, latent_channels, ch, groups):
        super().__init__()
        rev = list(ch)[::-1]
        self.conv_in = Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], None, "self", head_dim=rev[0], groups=groups, eps=1e-6)
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i in range(len(rev)):
            prev, out = out, rev[i]
            self.up_blocks.append(DecUpBlock(prev, out, i < len(rev) - 1, groups))
        self.conv_norm_out = GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_out = Conv2d(rev[-1], out_channels, 3, padding=1)


class _VaeConfig:
    scaling_factor = 0.18215


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4,
                 block_out_channels=(128, 256, 512, 512), groups=32):
        super().__init__()
        self.latent_channels = latent_channels
        self.config = _VaeConfig()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, groups)
        self.decoder = Decoder(out_channels, latent_channels, block_out_channels, groups)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = Conv2d(latent_channels, latent_channels, 1)
