"""Torch-tensor front end of the HIP kernels (device pointers + current HIP stream -> C ABI).

Activations are NHWC bf16 tensors ([N,H,W,C] or [rows, C]); C is always a multiple of 8 (thin tensors such as
images / latents are zero-padded to 8 channels).  PyTorch only owns memory and the stream here.
"""
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import capi
from .capi import (UR_ACT_GATE, UR_ACT_GEGLU, UR_ACT_GELU, UR_ACT_NONE, UR_ACT_RELU, UR_ACT_SILU, UR_ACT_TANH, ConvDesc,
                   check, lib)

BF16 = torch.bfloat16
_ws = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_SIDE = {}


def side_stream() -> "torch.cuda.Stream":
    """The per-device side stream on which independent branches of a forward run (parallel branches of the captured graph)."""
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev)
    return _SIDE[dev]


def workspace(dev, nbytes=192 << 20) -> torch.Tensor:
    """Split-K partial planes: one buffer for the side stream, one for everything else (kernels of the main branch - whatever
    stream or capture it runs under - are ordered among themselves; the side branch runs concurrently with them)."""
    cur = torch.cuda.current_stream()
    on_side = any(cur == s for s in _SIDE.values())
    key = (dev, "splitk", on_side)
    if key not in _ws or _ws[key].numel() * 4 < nbytes:
        _ws[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    return _ws[key]


def _gn_ws(dev, nbytes) -> torch.Tensor:
    key = (dev, "gn")
    if key not in _ws or _ws[key].numel() * 8 < nbytes:
        _ws[key] = torch.zeros(max(nbytes // 8 + 1, 1 << 17), dtype=torch.float64, device=dev)   # zero at rest
    return _ws[key]


def _gn_ab(dev, nbytes) -> torch.Tensor:
    key = (dev, "gn_ab")
    if key not in _ws or _ws[key].numel() * 4 < nbytes:
        _ws[key] = torch.empty(max(nbytes // 4 + 1, 1 << 18), dtype=torch.float32, device=dev)
    return _ws[key]


class StatsArena:
    """Bump allocator for the fp64 GroupNorm channel sums that conv epilogues produce (ur_conv_desc.gn_stats).
    One zero-fill of the used part per forward (`reset`) replaces a zero-fill per GroupNorm."""

    def __init__(self, dev, n_elems=48 << 20, dtype=torch.float64):
        self.buf = torch.zeros(n_elems, dtype=dtype, device=dev)
        self.off = 0
        self.high = 0

    def reset(self):
        if self.high:
            self.buf[:self.high].zero_()
        self.off = 0
        self._loop_high = 0

    def mark(self):
        return self.off

    def rewind(self, mark):
        """Reuse everything allocated since `mark` (one denoise step): re-zero that region and continue from the mark.
        Statistics produced before the mark (encoder skip features) stay alive."""
        hi = max(self.off, getattr(self, "_loop_high", 0))
        self._loop_high = hi
        if hi > mark:
            self.buf[mark:hi].zero_()
        self.off = mark

    def alloc(self, n):
        n = round_up(n, 2)
        assert self.off + n <= self.buf.numel(), "GroupNorm stats arena exhausted"
        v = self.buf[self.off:self.off + n]
        self.off += n
        self.high = max(self.high, self.off)
        return v


class _Arenas:
    """fp64 channel-sum arena (GroupNorm) + fp32 row-sum arena (LayerNorm fusion); reset / mark / rewind act on both."""

    def __init__(self, dev):
        self.gn = StatsArena(dev)
        self.rows = StatsArena(dev, 32 << 20, torch.float32)
        self.off = 0          # kept for callers that poke the GroupNorm offset directly

    def reset(self):
        self.gn.reset(); self.rows.reset()

    def mark(self):
        return (self.gn.mark(), self.rows.mark())

    def rewind(self, mark):
        self.gn.rewind(mark[0])
        self.rows.off = mark[1]          # row-sum partials are fully overwritten by their producers: no re-zeroing

    def alloc(self, n):
        return self.gn.alloc(n)


_arena = {}


def arena(dev=None) -> _Arenas:
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else dev
    if dev not in _arena:
        _arena[dev] = _Arenas(dev)
    return _arena[dev]


def ln_of(t):
    """Per-row (sum, sum of squares) attached to `t` by its producer GEMM (or None)."""
    return getattr(t, "_ln", None)


def gn_of(t):
    """Fused GroupNorm sums attached to tensor `t` by its producer (or None)."""
    return getattr(t, "_gn", None)


def carry(src, dst):
    """Propagate the producer's statistics across a reshape / view."""
    g = getattr(src, "_gn", None)
    if g is not None:
        dst._gn = g
    r = getattr(src, "_ln", None)
    if r is not None:
        dst._ln = r
    return dst


def round_up(v, m):
    return (v + m - 1) // m * m


# ------------------------------------------------------------------------------------------------ weights
@dataclass
class PackedConv:
    """bf16 [Cout][KH*KW*Cin] weight (K runs tap-major, channel-minor) + fp32 bias, padded for the kernel."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    cin: int          # padded input channels (multiple of 8)
    cout: int         # GEMM N (multiple of 4; for pair activations the interleaved a|g row count)
    cout_out: int     # channels actually produced (cout/2 for pair activations)
    k: int            # kernel size (1 or 3)
    groups: int = 1
    pair: bool = False
    ln_colsum: Optional[torch.Tensor] = None   # LayerNorm-fused GEMM: fp32 [cout] row sums of the folded bf16 weight
    ln_eps: float = 0.0
    kcm: bool = False  # K order (64-ch chunk, tap, ch) instead of (tap, ch): consecutive K tiles re-read the same pixels (L2)


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], dev, *, pair=False, groups=1, cin_pad=None, c1=None) -> PackedConv:
    """weight: [Cout, Cin/groups, k, k] (nn.Conv2d) or [N, K] (nn.Linear), fp32 master on any device.
    The repack (OIHW -> O,kh,kw,I; zero padding; a|g interleave; bf16 cast) runs on `dev` with torch copies."""
    w = weight.detach().to(dev, torch.float32)
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin_g, kh, kw = w.shape
    assert kh == kw and kh in (1, 3)
    cin_p = cin_pad or round_up(cin_g, 8)
    cout_p = round_up(cout, 8)           # outputs feed the next conv: keep C % 8 == 0 (padded rows are zero)
    if groups > 1:
        assert cin_g % 8 == 0 and (cout // groups) % 4 == 0
    if cin_p == cin_g and cout_p == cout:
        wp = w.permute(0, 2, 3, 1).contiguous()
    else:
        wp = torch.zeros(cout_p, kh, kw, cin_p, dtype=torch.float32, device=dev)
        wp[:cout, :, :, :cin_g] = w.permute(0, 2, 3, 1)
    b = None
    if bias is not None:
        b = torch.zeros(cout_p, dtype=torch.float32, device=dev)
        b[:cout] = bias.detach().to(dev, torch.float32)
    cout_out = cout_p
    if pair:
        half = cout // 2
        assert cout % 2 == 0 and half % 32 == 0, "pair activations need (Cout/2) % 32 == 0"
        idx = torch.arange(cout, device=dev).view(2, half // 32, 32).permute(1, 0, 2).reshape(-1)   # [blk][a|g][32]
        wp = wp[idx]
        b = b[idx] if b is not None else None
        cout_out = half
    # chunk-major K for 3x3 kernels: [Cout][kh*kw][Cin/64][64] -> [Cout][Cin/64][kh*kw][64]; c1 = channels of the first
    # source when the input is a virtual concat (chunks must not straddle it)
    kcm = kh == 3 and groups == 1 and cin_p % 64 == 0 and (c1 is None or c1 % 64 == 0) and os.environ.get("UR_KCM", "1") == "1"
    if kcm:
        wp = wp.reshape(cout_p, kh * kw, cin_p // 64, 64).permute(0, 2, 1, 3)
    return PackedConv(wp.reshape(cout_p, kh * kw * cin_p).to(BF16).contiguous(),
                      None if b is None else b.contiguous(), cin_p, cout_p, cout_out, kh, groups, pair, kcm=kcm)


# ------------------------------------------------------------------------------------------------ conv / gemm
def conv(x: torch.Tensor, pc: PackedConv, *, x2=None, residual=None, bias=None, act=UR_ACT_NONE, stride=1, pad=None,
         out_hw=None, upsample=False, out_f32=False, out_scale=1.0, out=None, yt=None, n_split=0, t_rows=0,
         colsum=None, colsum_scale=1.0, gn=False, rows=False, ln_stats=None):
    """x: [N,H,W,C1] bf16 (x2 optional [N,H,W,C2], virtual concat).  Returns [N,OH,OW,cout_out]."""
    assert x.dtype == BF16 and x.is_contiguous() and x.dim() == 4
    n, h, w_, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    g = pc.groups
    assert (c1 + c2) == pc.cin * g, f"Cin mismatch: {c1}+{c2} vs {pc.cin}*{g}"
    k = pc.k
    if pad is None:
        pad = (k // 2, k // 2)
    hin, win = (h * 2, w_ * 2) if upsample else (h, w_)
    if out_hw is None:
        out_hw = ((hin + 2 * pad[0] - k) // stride + 1, (win + 2 * pad[1] - k) // stride + 1)
    oh, ow = out_hw
    co_total = pc.cout_out
    if out is None and colsum is None:
        out = torch.empty((n, oh, ow, co_total), dtype=torch.float32 if out_f32 else BF16, device=x.device)
    d = ConvDesc()
    bias_t = pc.bias if bias is None else bias          # override: per-step (time-embedding) or per-image bias rows
    d.x, d.x2, d.w, d.bias = _ptr(x), _ptr(x2), _ptr(pc.w), _ptr(bias_t)
    if bias is not None and bias.dim() == 2 and bias.shape[0] > 1:
        d.bias_img_stride = bias.shape[1]
    d.residual, d.y, d.yt, d.colsum = _ptr(residual), _ptr(out), _ptr(yt), _ptr(colsum)
    stats = None
    if gn:      # the consumer of `out` is a GroupNorm: have the epilogue (or a fallback pass) leave its channel sums
        stats = arena(x.device).alloc(n * co_total * 2)
        d.gn_stats = stats.data_ptr()
    if ln_stats is not None:
        assert pc.ln_colsum is not None, "weights were not packed with pack_linear_ln"
        st, parts = ln_stats
        d.ln_stats, d.ln_colsum, d.ln_eps, d.ln_dim, d.ln_parts = st.data_ptr(), pc.ln_colsum.data_ptr(), pc.ln_eps, pc.cin, parts
    ws = workspace(x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.N, d.H, d.W = n, h, w_
    rstats = None
    d.C1, d.ldx, d.C2, d.ldx2 = (c1 // g if g > 1 else c1), c1, c2, c2
    d.Cout = pc.cout // g
    d.ldw = pc.w.shape[1]
    d.ldy = out.shape[-1] if out is not None else co_total
    d.ldr = residual.shape[-1] if residual is not None else 0
    d.KH = d.KW = k
    d.stride, d.pad_t, d.pad_l, d.OH, d.OW = stride, pad[0], pad[1], oh, ow
    d.upsample2x, d.act, d.out_f32 = int(upsample), act, int(out_f32)
    d.n_split, d.t_rows = n_split, t_rows
    d.k_chunk_major = int(pc.kcm and (c2 == 0 or c1 % 64 == 0))
    assert not pc.kcm or d.k_chunk_major, "chunk-major weights need a 64-aligned concat boundary"
    d.t_ld = yt.shape[-1] if yt is not None else 0
    d.out_scale, d.colsum_scale = out_scale, colsum_scale
    d.nbatch = g
    if g > 1:
        d.bs_x, d.bs_w, d.bs_bias, d.bs_y = c1 // g, (pc.cout // g) * pc.w.shape[1], pc.cout // g, pc.cout_out // g
        d.bs_r = pc.cout_out // g
    if rows:    # the consumer of `out` is a LayerNorm-fused GEMM: leave per-row (sum, sumsq) partials, one plane per N tile
        parts = lib.ur_conv2d_row_stat_parts(d)
        if parts <= 0:
            check(parts if parts < 0 else -1)
        rstats = (arena(x.device).rows.alloc(parts * n * oh * ow * 2), parts)
        d.row_stats = rstats[0].data_ptr()
    check(lib.ur_conv2d_nhwc(d, _stream()))
    if stats is not None:
        out._gn = stats
    if rstats is not None:
        out._ln = rstats
    return out


def pack_linear_ln(weight, bias, gamma, beta, eps, dev, *, pair=False) -> PackedConv:
    """Linear(LayerNorm(x)) folded for the LN-fused GEMM epilogue: w' = W*gamma (bf16), bias' = W.beta + b,
    ln_colsum[n] = sum_k bf16(w'[n,k]).  The kernel computes rstd*(w'.x - mean*ln_colsum) + bias'."""
    w = weight.detach().to(dev, torch.float32)
    g, b0 = gamma.detach().to(dev, torch.float32), beta.detach().to(dev, torch.float32)
    wf = w * g[None, :]
    t = w @ b0 + (bias.detach().to(dev, torch.float32) if bias is not None else 0.0)
    pc = pack_conv(wf, t, dev, pair=pair)
    pc.ln_colsum = pc.w.float().sum(dim=1).contiguous()          # in packed (possibly a|g interleaved) row order
    pc.ln_eps = float(eps)
    return pc


def linear(x: torch.Tensor, pc: PackedConv, **kw):
    """x: [..., K] bf16 -> [..., N]; runs as a 1x1 conv over a [1,1,rows,K] image."""
    shp = x.shape
    rows = x.numel() // shp[-1]
    res = kw.pop("residual", None)
    if res is not None:
        res = res.reshape(1, 1, rows, res.shape[-1])
    gn = kw.pop("gn", False)
    gn_hw = kw.pop("gn_hw", None)       # (N, HW): how the rows split into images for the fused GroupNorm sums
    if kw.get("ln_stats") is None and getattr(pc, "ln_colsum", None) is not None:
        raise ValueError("LayerNorm-folded weights need ln_stats")
    if gn:
        n_img, hw = gn_hw
        y = conv(x.reshape(n_img, 1, hw, shp[-1]), pc, residual=None if res is None else res.reshape(n_img, 1, hw, -1), gn=True, **kw)
    else:
        y = conv(x.reshape(1, 1, rows, shp[-1]), pc, residual=res, **kw)
    return None if y is None else carry(y, y.reshape(*shp[:-1], y.shape[-1]))


def bmm_nt(a: torch.Tensor, bmat: torch.Tensor, *, out_f32=False, out_scale=1.0):
    """Batched C[b] = A[b] @ B[b]^T with A:[B,M,K], B:[B,N,K] bf16 (last dim contiguous; row strides free)."""
    B, M, K = a.shape
    N = bmat.shape[1]
    assert a.stride(2) == 1 and bmat.stride(2) == 1 and K % 8 == 0 and N % 4 == 0
    out = torch.empty((B, M, N), dtype=torch.float32 if out_f32 else BF16, device=a.device)
    d = ConvDesc()
    d.x, d.w, d.y = a.data_ptr(), bmat.data_ptr(), out.data_ptr()
    ws = workspace(a.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.N, d.H, d.W, d.C1, d.ldx, d.Cout, d.ldw, d.ldy = 1, 1, M, K, a.stride(1), N, bmat.stride(1), N
    d.KH = d.KW = 1
    d.stride, d.OH, d.OW, d.out_f32, d.out_scale, d.nbatch = 1, 1, M, int(out_f32), out_scale, B
    d.bs_x, d.bs_w, d.bs_y = a.stride(0), bmat.stride(0), M * N
    check(lib.ur_conv2d_nhwc(d, _stream()))
    return out


# ------------------------------------------------------------------------------------------------ norms
def group_norm(x: torch.Tensor, gamma, beta, groups: int, eps: float, silu=False, x2=None, use_pre=True):
    """x: [N,H,W,C1] bf16 (+ optional x2 [N,H,W,C2], normalised as one concatenated tensor) -> [N,H,W,C1+C2].
    gamma/beta fp32 [C] or None (InstanceNorm when groups == C)."""
    assert x.dtype == BF16 and x.is_contiguous()
    n, c1 = x.shape[0], x.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    hw = x.numel() // (n * c1)
    out = torch.empty((*x.shape[:-1], c1 + c2), dtype=BF16, device=x.device)
    ws = _gn_ws(x.device, lib.ur_groupnorm_ws_bytes(n, c1 + c2))
    ab = _gn_ab(x.device, lib.ur_groupnorm_ab_bytes(n, c1 + c2))
    pre1 = gn_of(x) if use_pre else None
    pre2 = gn_of(x2) if (use_pre and x2 is not None) else None
    check(lib.ur_groupnorm_nhwc(x.data_ptr(), _ptr(x2), out.data_ptr(), _ptr(gamma), _ptr(beta), n, hw, c1, c2, groups, eps,
                                int(silu), ws.data_ptr(), ab.data_ptr(), _ptr(pre1), _ptr(pre2), _stream()))
    return out


def layer_norm(x: torch.Tensor, gamma, beta, eps: float):
    assert x.dtype == BF16 and x.is_contiguous()
    c = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.ur_layernorm_rows(x.data_ptr(), out.data_ptr(), _ptr(gamma), _ptr(beta), x.numel() // c, c, eps, _stream()))
    return out


def softmax_rows(s: torch.Tensor, ldp=None):
    """s: [..., cols] fp32 -> bf16 probabilities [..., ldp] (columns >= cols are zero)."""
    cols = s.shape[-1]
    ldp = ldp or round_up(cols, 8)
    p = torch.empty((*s.shape[:-1], ldp), dtype=BF16, device=s.device)
    check(lib.ur_softmax_rows_f32(s.data_ptr(), p.data_ptr(), s.numel() // cols, cols, ldp, _stream()))
    return p


# ------------------------------------------------------------------------------------------------ attention
def attention(q, k, vt, heads: int, head_dim: int, tq: int, tk: int, scale: float, *, ldq, ldk, bs_q, bs_k, bs_vt,
              batch: int, out=None):
    """q:[B,Tq,ldq] k:[B?,Tk,ldk] vt:[B?,H*D,ldvt] (raw tensors; strides given explicitly) -> o [B,Tq,H*D]."""
    c = heads * head_dim
    out = torch.empty((batch, tq, c), dtype=BF16, device=q.device) if out is None else out
    check(lib.ur_attention_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), batch, heads, tq, tk, head_dim,
                               ldq, ldk, vt.shape[-1], c, bs_q, bs_k, bs_vt, tq * c, scale, _stream()))
    return out


# ------------------------------------------------------------------------------------------------ misc
def dwconv3x3(x, w9c, bias, gate=False):
    n, h, w_, c = x.shape
    out = torch.empty((n, h, w_, c // 2 if gate else c), dtype=BF16, device=x.device)
    check(lib.ur_dwconv3x3_nhwc(x.data_ptr(), w9c.data_ptr(), bias.data_ptr(), out.data_ptr(), n, h, w_, c, int(gate), _stream()))
    return out


def avgpool(x):
    n, c = x.shape[0], x.shape[-1]
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(lib.ur_avgpool_hw(x.data_ptr(), out.data_ptr(), n, x.numel() // (n * c), c, _stream()))
    return out


def scale_channels(x, s, residual=None):
    n, c = x.shape[0], x.shape[-1]
    out = torch.empty_like(x)
    check(lib.ur_scale_channels(x.data_ptr(), s.data_ptr(), _ptr(residual), out.data_ptr(), n, x.numel() // (n * c), c, _stream()))
    return out


def spade_modulate(n, gb, residual=None):
    """y = n * (1 + gamma) + beta (+ residual); gb [..., 2C] = gamma | beta (spade.py:69)."""
    c = n.shape[-1]
    out = torch.empty_like(n)
    check(lib.ur_spade_modulate(n.data_ptr(), gb.data_ptr(), gb.shape[-1], _ptr(residual), out.data_ptr(), n.numel() // c, c, _stream()))
    return out


def axpy_channels(a, b, s):
    c = a.shape[-1]
    out = torch.empty_like(a)
    check(lib.ur_axpy_channels(a.data_ptr(), b.data_ptr(), s.data_ptr(), out.data_ptr(), a.numel() // c, c, _stream()))
    return out


def linear_f32(x, w, bias, act=UR_ACT_NONE, groups=1):
    """x [M,K] fp32, w [N,K/groups] fp32 -> [M,N] fp32."""
    m, k = x.shape
    n = w.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(lib.ur_linear_f32(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), m, n, k, groups, act, _stream()))
    return out


def tfa_prompt_update(pooled, cond):
    b, t, d = cond.shape
    upd = torch.empty_like(cond)
    check(lib.ur_tfa_prompt_update(pooled.data_ptr(), cond.data_ptr(), upd.data_ptr(), b, t, d, _stream()))
    return upd


def vec_mul_group(a, b, groups):
    out = torch.empty_like(a)
    check(lib.ur_vec_mul_group(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1], groups, _stream()))
    return out


def nchw_to_nhwc(x: torch.Tensor, cpad=None, image=False):
    """fp32 NCHW -> bf16 NHWC (channels zero-padded to a multiple of 8); image=True applies x*2-1."""
    x = x.contiguous().float()
    n, c, h, w_ = x.shape
    cpad = cpad or round_up(c, 8)
    out = torch.empty((n, h, w_, cpad), dtype=BF16, device=x.device)
    fn = lib.ur_image_to_nhwc if image else lib.ur_nchw_f32_to_nhwc
    check(fn(x.data_ptr(), out.data_ptr(), n, c, h, w_, cpad, _stream()))
    return out


def image_resize_pad(img: torch.Tensor, rh: int, rw: int, ph: int, pw: int, mul=2.0, add=-1.0, cpad=None):
    """DiffUIE.forward pre-processing (unifie.py:124-134) + x*2-1 + layout: fp32 NCHW -> bf16 NHWC [N, rh+ph, rw+pw, cpad]."""
    img = img.contiguous().float()
    n, c, h, w_ = img.shape
    cpad = cpad or round_up(c, 8)
    out = torch.empty((n, rh + ph, rw + pw, cpad), dtype=BF16, device=img.device)
    check(lib.ur_image_resize_pad_nhwc(img.data_ptr(), out.data_ptr(), n, c, h, w_, rh, rw, ph, pw, cpad, mul, add, _stream()))
    return out


def image_unpad_resize(x: torch.Tensor, c: int, crop_hw, out_hw, mul=1.0, add=0.0, quantize=False):
    """Post-processing (unifie.py:164-168 [+ eval_image_restoration.py:71 when quantize]): NHWC -> crop -> bicubic -> fp32 NCHW."""
    n, xh, xw, ld = x.shape
    out = torch.empty((n, c, out_hw[0], out_hw[1]), dtype=torch.float32, device=x.device)
    check(lib.ur_image_unpad_resize_nchw(x.data_ptr(), int(x.dtype == torch.float32), out.data_ptr(), n, c, xh, xw, ld, crop_hw[0],
                                         crop_hw[1], out_hw[0], out_hw[1], mul, add, int(quantize), _stream()))
    return out


def nhwc_to_nchw(x: torch.Tensor, c=None, mul=1.0, add=0.0):
    n, h, w_, ld = x.shape
    c = c or ld
    out = torch.empty((n, c, h, w_), dtype=torch.float32, device=x.device)
    check(lib.ur_nhwc_to_nchw_f32(x.data_ptr(), int(x.dtype == torch.float32), out.data_ptr(), n, c, h, w_, ld, mul, add, _stream()))
    return out


def vae_sample(moments_f32, noise_nchw, clat, scale):
    n, h, w_, ld = moments_f32.shape
    z = torch.empty((n, h, w_, 8), dtype=torch.float32, device=moments_f32.device)
    zb = torch.empty((n, h, w_, 8), dtype=BF16, device=moments_f32.device)
    check(lib.ur_vae_sample(moments_f32.data_ptr(), ld, noise_nchw.data_ptr(), z.data_ptr(), zb.data_ptr(), n, h * w_, clat, 8,
                            scale, _stream()))
    return z, zb


def add_noise(z0, noise_nchw, clat, sa, sb):
    n, h, w_, cp = z0.shape
    zt, zb = torch.empty_like(z0), torch.empty(z0.shape, dtype=BF16, device=z0.device)
    check(lib.ur_add_noise(z0.data_ptr(), noise_nchw.data_ptr(), zt.data_ptr(), zb.data_ptr(), n, h * w_, clat, cp, sa, sb, _stream()))
    return zt, zb


def ddim_step_(zt, zt_bf16, eps_f32, clat, c_x, c_e):
    cp = zt.shape[-1]
    check(lib.ur_ddim_step(zt.data_ptr(), eps_f32.data_ptr(), eps_f32.shape[-1], zt_bf16.data_ptr(), zt.numel() // cp, clat, cp,
                           c_x, c_e, _stream()))


def f32_to_bf16(x, c, mul=1.0, cpad=8):
    ld = x.shape[-1]
    out = torch.empty((*x.shape[:-1], cpad), dtype=BF16, device=x.device)
    check(lib.ur_f32_to_bf16_scaled(x.data_ptr(), ld, out.data_ptr(), x.numel() // ld, c, cpad, mul, _stream()))
    return out


def profile_enable(on: bool):
    check(lib.ur_profile_enable(int(on)))


def profile_report() -> dict:
    import ctypes
    import json
    buf = ctypes.create_string_buffer(1 << 20)
    check(lib.ur_profile_report(buf, len(buf)))
    return json.loads(buf.value.decode())
