"""Torch-tensor front end of the HIP kernels (device pointers + current HIP stream -> C ABI).

Activations are NHWC bf16 tensors ([N,H,W,C] or [rows, C]); C is always a multiple of 8 (thin tensors such as
images / latents are zero-padded to 8 channels).  PyTorch only owns memory and the stream here.
"""
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import capi
from .capi import (UR_ACT_GATE, UR_ACT_GEGLU, UR_ACT_GELU, UR_ACT_NONE, UR_ACT_RELU, UR_ACT_SILU, UR_ACT_TANH, UR_DT_BF16,
                   UR_DT_F16, ConvDesc, ConvPlan, check, lib)

BF16 = torch.bfloat16
F16 = torch.float16
_ws = {}

# ---- compute dtype: the 16-bit type activations / weights are stored in and fed to the matrix cores ----------------------
_DTYPES = {"bf16": BF16, "bfloat16": BF16, BF16: BF16, "fp16": F16, "float16": F16, "f16": F16, "16": F16, F16: F16}
_act = BF16


def set_dtype(dt) -> torch.dtype:
    """Select the 16-bit activation / weight type ("bf16" | "fp16") for subsequent ops; returns the torch dtype."""
    global _act
    if dt not in _DTYPES:
        raise ValueError(f"unsupported compute dtype {dt!r}: choose 'bf16' or 'fp16'")
    _act = _DTYPES[dt]
    return _act


def act_dtype() -> torch.dtype:
    return _act


def _dt(t: Optional[torch.Tensor] = None) -> int:
    """UR_DT_* code of tensor `t` (or of the current compute dtype)."""
    d = _act if t is None else t.dtype
    if d == BF16:
        return UR_DT_BF16
    if d == F16:
        return UR_DT_F16
    raise TypeError(f"expected a bf16 / fp16 tensor, got {d}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def workspace(dev, nbytes=192 << 20) -> torch.Tensor:
    """Split-K partial planes: ONE fixed-size buffer per device (every launch of a forward is ordered on one stream).  Allocated
    once and never regrown (captured graphs keep its address); the library falls back to fewer splits when a launch would not fit."""
    key = (dev, "splitk")
    if key not in _ws:
        _ws[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    return _ws[key]


def ln_of(t):
    """Per-row (sum, sum of squares) attached to `t` by its producer GEMM (or None)."""
    return getattr(t, "_ln", None)


def gn_of(t):
    """(partial plane fp32 [N][P][C][2], P) attached to tensor `t` by its producer (or None): its GroupNorm statistics."""
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
# A/B (UR_WEIGHT_ARENA=<GiB per block>): packed weights live in a few multi-GiB allocations instead of one allocation each - one UNet
# step walks 1.7 GB of weights, and how many address translations that costs depends on how the driver can map them.
_ARENA_GB = float(os.environ.get("UR_WEIGHT_ARENA", "0") or 0)
_arenas = {}


def _arena_place(t: torch.Tensor) -> torch.Tensor:
    if _ARENA_GB <= 0 or not t.is_cuda:
        return t
    nbytes = round_up(t.numel() * t.element_size(), 4096)
    blocks = _arenas.setdefault(t.device, [])
    if not blocks or blocks[-1][1] + nbytes > blocks[-1][0].numel():
        blocks.append([torch.empty(max(int(_ARENA_GB * (1 << 30)), nbytes), dtype=torch.uint8, device=t.device), 0])
    buf, off = blocks[-1]
    blocks[-1][1] = off + nbytes
    v = buf[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
    v.copy_(t)
    return v


@dataclass
class PackedConv:
    """16-bit [Cout][KH*KW*Cin] weight (K runs tap-major, channel-minor) + fp32 bias, padded for the kernel."""
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
    w_frag: Optional[torch.Tensor] = None      # MFMA-fragment-major copy for the weight-streaming kernel (built on first use)

    def frag(self) -> torch.Tensor:
        """ur_conv_desc.w_frag: [Cout/128][Cin/64][tap][k-step][row block][lane][8] from the chunk-major rows (csrc/conv_wstream.hip)."""
        if self.w_frag is None:
            nt, nc = self.cout // 128, self.cin // 64
            w = self.w.view(nt, 4, 32, nc, 9, 4, 2, 8)                    # [nt][row block][row][chunk][tap][k-step][half][8]
            self.w_frag = _arena_place(w.permute(0, 3, 4, 5, 1, 6, 2, 7).contiguous())  # lane = half * 32 + row
        return self.w_frag


def wants_frag(pc: "PackedConv", n, h, w_, c1, c2, stride, upsample, groups) -> bool:
    """The launches csrc/igemm.hip sends to the weight-streaming kernel (it also checks): 3x3 on 8 x 8 maps of <= 16 images."""
    return (pc.k == 3 and pc.kcm and groups == 1 and stride == 1 and not upsample and h == 8 and w_ == 8 and n <= 16 and not pc.pair
            and pc.cout % 128 == 0 and (c1 + c2) % 256 == 0 and (c1 + c2) >= 512 and (c2 == 0 or c1 % 64 == 0)
            and os.environ.get("UR_IGEMM_NOWSTREAM") is None)


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], dev, *, pair=False, groups=1, cin_pad=None, c1=None, group_halo=False) -> PackedConv:
    """weight: [Cout, Cin/groups, k, k] (nn.Conv2d) or [N, K] (nn.Linear), fp32 master on any device.
    The repack (OIHW -> O,kh,kw,I; zero padding; a|g interleave; bf16 cast) runs on `dev` with torch copies.
    group_halo: a grouped 3x3 conv whose groups are halo-kernel sized (>= 64 channels in, a multiple of 128 out) - chunk-major
    weights, one halo launch per group on its channel slice (csrc/igemm.hip conv_impl)."""
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
        if cout % 2 or half % 32:
            raise NotImplementedError(f"UR_E_UNSUPPORTED: pair activations (GEGLU / SimpleGate) need (Cout/2) % 32 == 0, got {half}")
        idx = torch.arange(cout, device=dev).view(2, half // 32, 32).permute(1, 0, 2).reshape(-1)   # [blk][a|g][32]
        wp = wp[idx]
        b = b[idx] if b is not None else None
        cout_out = half
    # chunk-major K for 3x3 kernels: [Cout][kh*kw][Cin/64][64] -> [Cout][Cin/64][kh*kw][64]; c1 = channels of the first
    # source when the input is a virtual concat (chunks must not straddle it)
    # (grouped convs too when every group is chunk-sized: each group then runs the halo kernel on its channel slice)
    kcm = kh == 3 and (groups == 1 or (group_halo and cout_p == cout)) and cin_p % 64 == 0 and (c1 is None or c1 % 64 == 0) and os.environ.get("UR_KCM", "1") == "1"
    if kcm:
        wp = wp.reshape(cout_p, kh * kw, cin_p // 64, 64).permute(0, 2, 1, 3)
    return PackedConv(_arena_place(wp.reshape(cout_p, kh * kw * cin_p).to(_act).contiguous()),
                      None if b is None else b.contiguous(), cin_p, cout_p, cout_out, kh, groups, pair, kcm=kcm)


# ------------------------------------------------------------------------------------------------ conv / gemm
def conv(x: torch.Tensor, pc: PackedConv, *, x2=None, residual=None, bias=None, act=UR_ACT_NONE, stride=1, pad=None,
         out_hw=None, upsample=False, out_f32=False, out_scale=1.0, out=None, yt=None, n_split=0, t_rows=0,
         gn=False, store=True, rows=False, ln_stats=None, gn_ab=None, gn_silu=False):
    """x: [N,H,W,C1] 16-bit (x2 optional [N,H,W,C2], virtual concat).  Returns [N,OH,OW,cout_out].
    gn=True: the consumer of the output is a GroupNorm / InstanceNorm / global average pool - leave its partial statistics
    (out._gn); store=False (with gn): only the statistics are wanted, the output tensor is not written (returns the plane).
    rows=True: leave per-row sums for a LayerNorm-folded consumer GEMM (out._ln).
    gn_ab / gn_silu: read act(a*x+b) instead of x (GroupNorm apply fused into the loader; only where conv_plan says so)."""
    if x.dtype not in (BF16, F16) or x.dtype != pc.w.dtype or not x.is_contiguous() or x.dim() != 4:
        raise ValueError(f"conv: x must be a contiguous 4-d {pc.w.dtype} tensor, got {x.dtype} {tuple(x.shape)}")
    n, h, w_, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    g = pc.groups
    if (c1 + c2) != pc.cin * g:
        raise ValueError(f"conv: Cin mismatch: {c1}+{c2} vs {pc.cin}*{g}")
    k = pc.k
    if pad is None:
        pad = (k // 2, k // 2)
    hin, win = (h * 2, w_ * 2) if upsample else (h, w_)
    if out_hw is None:
        out_hw = ((hin + 2 * pad[0] - k) // stride + 1, (win + 2 * pad[1] - k) // stride + 1)
    oh, ow = out_hw
    co_total = pc.cout_out
    if out is None and store:
        out = torch.empty((n, oh, ow, co_total), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    d = ConvDesc()
    d.dtype = _dt(x)
    bias_t = pc.bias if bias is None else bias          # override: per-step (time-embedding) or per-image bias rows
    d.x, d.x2, d.w, d.bias = _ptr(x), _ptr(x2), _ptr(pc.w), _ptr(bias_t)
    if bias is not None and bias.dim() == 2 and bias.shape[0] > 1:
        d.bias_img_stride = bias.shape[1]
    d.residual, d.y, d.yt = _ptr(residual), _ptr(out), _ptr(yt)
    if ln_stats is not None:
        assert pc.ln_colsum is not None, "weights were not packed with pack_linear_ln"
        st, parts = ln_stats
        d.ln_stats, d.ln_colsum, d.ln_eps, d.ln_dim, d.ln_parts = st.data_ptr(), pc.ln_colsum.data_ptr(), pc.ln_eps, pc.cin, parts
    if gn_ab is not None:
        d.gn_ab, d.gn_silu = gn_ab.data_ptr(), int(gn_silu)
    ws = workspace(x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.N, d.H, d.W = n, h, w_
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
    if wants_frag(pc, n, h, w_, c1, c2, stride, upsample, g) and pad == (1, 1):
        d.w_frag = pc.frag().data_ptr()
    d.t_ld = yt.shape[-1] if yt is not None else 0
    d.out_scale = out_scale
    d.nbatch = g
    if g > 1:
        d.bs_x, d.bs_w, d.bs_bias, d.bs_y = c1 // g, (pc.cout // g) * pc.w.shape[1], pc.cout // g, pc.cout_out // g
        d.bs_r = pc.cout_out // g
    stats = rstats = None
    if gn or rows:      # the consumer is a norm: plan the launch, then hand it the planes to fill (plain stores, no atomics)
        if gn:
            d.gn_part = 16      # dummy non-null values: the plan must see what the real launch will see
        if rows:
            d.row_stats = 16
        plan = ConvPlan()
        check(lib.ur_conv2d_plan(d, plan))
        if gn:
            if not store and not plan.gn_fused:
                raise NotImplementedError("store=False needs a launch whose epilogue writes the statistics (conv_plan().gn_fused)")
            stats = (torch.empty((n, plan.gn_parts, co_total, 2), dtype=torch.float32, device=x.device), plan.gn_parts)
            d.gn_part = stats[0].data_ptr()
        if rows:
            rstats = (torch.empty((plan.row_stat_parts, n * oh * ow, 2), dtype=torch.float32, device=x.device), plan.row_stat_parts)
            d.row_stats = rstats[0].data_ptr()
    check(lib.ur_conv2d_nhwc(d, _stream()))
    if not store:
        return stats
    if stats is not None:
        out._gn = stats
    if rstats is not None:
        out._ln = rstats
    return out


def conv_plan(x: torch.Tensor, pc: PackedConv, *, x2=None, stride=1, pad=None, upsample=False, gn=False, gn_ab=False,
              act=UR_ACT_NONE, residual=False, store=True) -> ConvPlan:
    """What `conv` would do for this (shape, weights) - used to decide whether the GroupNorm apply can ride in the conv."""
    n, h, w_, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    k, g = pc.k, pc.groups
    pad = (k // 2, k // 2) if pad is None else pad
    hin, win = (h * 2, w_ * 2) if upsample else (h, w_)
    oh, ow = (hin + 2 * pad[0] - k) // stride + 1, (win + 2 * pad[1] - k) // stride + 1
    d = ConvDesc()
    d.dtype = _dt(x)
    d.x, d.w, d.y = 16, 16, (16 if store else None)
    d.x2 = 16 if c2 else None
    d.residual = 16 if residual else None
    d.gn_part = 16 if gn else None
    d.gn_ab = 16 if gn_ab else None
    ws = workspace(x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.N, d.H, d.W = n, h, w_
    d.C1, d.ldx, d.C2, d.ldx2 = (c1 // g if g > 1 else c1), c1, c2, c2
    d.Cout, d.ldw, d.ldy, d.ldr = pc.cout // g, pc.w.shape[1], pc.cout_out, pc.cout_out if residual else 0
    d.KH = d.KW = k
    d.stride, d.pad_t, d.pad_l, d.OH, d.OW = stride, pad[0], pad[1], oh, ow
    d.upsample2x, d.act, d.out_scale, d.nbatch = int(upsample), act, 1.0, g
    d.k_chunk_major = int(pc.kcm and (c2 == 0 or c1 % 64 == 0))
    if wants_frag(pc, n, h, w_, c1, c2, stride, upsample, g) and pad == (1, 1):
        d.w_frag = 16
    if g > 1:
        d.bs_x, d.bs_w, d.bs_bias, d.bs_y, d.bs_r = c1 // g, (pc.cout // g) * pc.w.shape[1], pc.cout // g, pc.cout_out // g, pc.cout_out // g
    plan = ConvPlan()
    check(lib.ur_conv2d_plan(d, plan))
    return plan


def pack_linear_ln(weight, bias, gamma, beta, eps, dev, *, pair=False) -> PackedConv:
    """Linear(LayerNorm(x)) folded for the LN-fused GEMM epilogue: w' = W*gamma (bf16), bias' = W.beta + b,
    ln_colsum[n] = sum_k bf16(w'[n,k]).  The kernel computes rstd*(w'.x - mean*ln_colsum) + bias'."""
    w = weight.detach().to(dev, torch.float32)
    g, b0 = gamma.detach().to(dev, torch.float32), beta.detach().to(dev, torch.float32)
    wf = w * g[None, :]
    t = w @ b0 + (bias.detach().to(dev, torch.float32) if bias is not None else 0.0)
    pc = pack_conv(wf, t, dev, pair=pair)
    pc.ln_colsum = pc.w.float().sum(dim=1).contiguous()          # of the ROUNDED weights, in packed (possibly a|g interleaved) row order
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
    if a.dtype != bmat.dtype or a.dtype not in (BF16, F16):
        raise ValueError(f"bmm_nt: both operands must share one 16-bit type, got {a.dtype} and {bmat.dtype}")
    assert a.stride(2) == 1 and bmat.stride(2) == 1 and K % 8 == 0 and N % 4 == 0
    out = torch.empty((B, M, N), dtype=torch.float32 if out_f32 else a.dtype, device=a.device)
    d = ConvDesc()
    d.dtype = _dt(a)
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
def gn_partials(x: torch.Tensor):
    """(plane, P) of x: the producer's, or from a statistics pass over x (deterministic partial planes, no atomics)."""
    pre = gn_of(x)
    if pre is not None:
        return pre
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    parts = lib.ur_groupnorm_stats_parts(n, hw, c)
    if parts <= 0:
        check(parts if parts < 0 else -1)
    plane = torch.empty((n, parts, c, 2), dtype=torch.float32, device=x.device)
    check(lib.ur_groupnorm_stats(x.data_ptr(), plane.data_ptr(), n, hw, c, _dt(x), _stream()))
    return plane, parts


def gn_finalize(x: torch.Tensor, gamma, beta, groups: int, eps: float, x2=None, use_pre=True, want_mean=False):
    """GroupNorm statistics of x (| x2, virtually concatenated) -> fp32 ab [N][2][C] with GroupNorm(x) = a*x + b
    (want_mean: the [N][groups] group means instead)."""
    n, c1 = x.shape[0], x.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    hw = x.numel() // (n * c1)
    if not use_pre:
        x = x.view(x.shape)          # fresh tensor object: no producer attributes
        x2 = None if x2 is None else x2.view(x2.shape)
    p1, n1 = gn_partials(x)
    p2, n2 = gn_partials(x2) if x2 is not None else (None, 0)
    out = torch.empty((n, groups) if want_mean else (n, 2, c1 + c2), dtype=torch.float32, device=x.device)
    check(lib.ur_groupnorm_finalize(p1.data_ptr(), n1, c1, _ptr(p2), n2, c2, _ptr(gamma), _ptr(beta), n, hw, groups, eps,
                                    None if want_mean else out.data_ptr(), out.data_ptr() if want_mean else None, _stream()))
    return out


def gn_finalize_planes(plane: torch.Tensor, parts: int, hw: int):
    """Channel means [N][C] straight from a partial plane [N][P][C][2] (statistics-only conv launch)."""
    n, c = plane.shape[0], plane.shape[2]
    out = torch.empty((n, c), dtype=torch.float32, device=plane.device)
    check(lib.ur_groupnorm_finalize(plane.data_ptr(), parts, c, None, 0, 0, None, None, n, hw, c, 0.0, None, out.data_ptr(), _stream()))
    return out


def gn_apply(x: torch.Tensor, ab: torch.Tensor, silu=False, x2=None):
    n, c1 = x.shape[0], x.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    hw = x.numel() // (n * c1)
    out = torch.empty((*x.shape[:-1], c1 + c2), dtype=x.dtype, device=x.device)
    check(lib.ur_groupnorm_apply_act(x.data_ptr(), _ptr(x2), out.data_ptr(), ab.data_ptr(), n, hw, c1, c2, int(silu), _dt(x), _stream()))
    return out


def group_norm(x: torch.Tensor, gamma, beta, groups: int, eps: float, silu=False, x2=None, use_pre=True):
    """x: [N,H,W,C1] 16-bit (+ optional x2 [N,H,W,C2], normalised as one concatenated tensor) -> [N,H,W,C1+C2].
    gamma/beta fp32 [C] or None (InstanceNorm when groups == C)."""
    assert x.dtype in (BF16, F16) and x.is_contiguous()
    return gn_apply(x, gn_finalize(x, gamma, beta, groups, eps, x2=x2, use_pre=use_pre), silu, x2=x2)


def layer_norm(x: torch.Tensor, gamma, beta, eps: float):
    assert x.dtype in (BF16, F16) and x.is_contiguous()
    c = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.ur_layernorm_rows(x.data_ptr(), out.data_ptr(), _ptr(gamma), _ptr(beta), x.numel() // c, c, eps, _dt(x), _stream()))
    return out


def softmax_rows(s: torch.Tensor, ldp=None):
    """s: [..., cols] fp32 -> 16-bit probabilities [..., ldp] (columns >= cols are zero)."""
    cols = s.shape[-1]
    ldp = ldp or round_up(cols, 8)
    p = torch.empty((*s.shape[:-1], ldp), dtype=_act, device=s.device)
    check(lib.ur_softmax_rows_f32(s.data_ptr(), p.data_ptr(), s.numel() // cols, cols, ldp, _dt(), _stream()))
    return p


# ------------------------------------------------------------------------------------------------ attention
def attention(q, k, vt, heads: int, head_dim: int, tq: int, tk: int, scale: float, *, ldq, ldk, bs_q, bs_k, bs_vt,
              batch: int, out=None):
    """q:[B,Tq,ldq] k:[B?,Tk,ldk] vt:[B?,H*D,ldvt] (raw tensors; strides given explicitly) -> o [B,Tq,H*D]."""
    c = heads * head_dim
    if not (q.dtype == k.dtype == vt.dtype) or q.dtype not in (BF16, F16):
        raise ValueError(f"attention: q, k, v^T must share one 16-bit type, got {q.dtype}, {k.dtype}, {vt.dtype}")
    out = torch.empty((batch, tq, c), dtype=q.dtype, device=q.device) if out is None else out
    nws = lib.ur_attention_workspace_bytes(batch, heads, tq, tk, head_dim)       # key-split last round of the d = 64 kernel (0: none)
    ws = torch.empty(nws, dtype=torch.uint8, device=q.device) if nws else None
    check(lib.ur_attention_fwd_ws(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), batch, heads, tq, tk, head_dim,
                                  ldq, ldk, vt.shape[-1], c, bs_q, bs_k, bs_vt, tq * c, scale, _ptr(ws), nws, _dt(q), _stream()))
    return out


# ------------------------------------------------------------------------------------------------ misc
def dwconv3x3(x, w9c, bias, gate=False):
    n, h, w_, c = x.shape
    out = torch.empty((n, h, w_, c // 2 if gate else c), dtype=x.dtype, device=x.device)
    check(lib.ur_dwconv3x3_nhwc(x.data_ptr(), w9c.data_ptr(), bias.data_ptr(), out.data_ptr(), n, h, w_, c, int(gate), _dt(x), _stream()))
    return out


def avgpool(x):
    """Mean over HW -> fp32 [N][C]: a finalize over the tensor's (producer-side or freshly computed) partial sums."""
    return gn_finalize(x, None, None, x.shape[-1], 0.0, want_mean=True)


def scale_channels(x, s, residual=None):
    n, c = x.shape[0], x.shape[-1]
    out = torch.empty_like(x)
    check(lib.ur_scale_channels(x.data_ptr(), s.data_ptr(), _ptr(residual), out.data_ptr(), n, x.numel() // (n * c), c, _dt(x), _stream()))
    return out


def spade_modulate(n, gb, residual=None):
    """y = n * (1 + gamma) + beta (+ residual); gb [..., 2C] = gamma | beta (spade.py:69)."""
    c = n.shape[-1]
    out = torch.empty_like(n)
    check(lib.ur_spade_modulate(n.data_ptr(), gb.data_ptr(), gb.shape[-1], _ptr(residual), out.data_ptr(), n.numel() // c, c, _dt(n), _stream()))
    return out


def axpy_channels(a, b, s):
    c = a.shape[-1]
    out = torch.empty_like(a)
    check(lib.ur_axpy_channels(a.data_ptr(), b.data_ptr(), s.data_ptr(), out.data_ptr(), a.numel() // c, c, _dt(a), _stream()))
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
    """fp32 NCHW -> 16-bit NHWC (channels zero-padded to a multiple of 8); image=True applies x*2-1."""
    x = x.contiguous().float()
    n, c, h, w_ = x.shape
    cpad = cpad or round_up(c, 8)
    out = torch.empty((n, h, w_, cpad), dtype=_act, device=x.device)
    fn = lib.ur_image_to_nhwc if image else lib.ur_nchw_f32_to_nhwc
    check(fn(x.data_ptr(), out.data_ptr(), n, c, h, w_, cpad, _dt(), _stream()))
    return out


def image_resize_pad(img: torch.Tensor, rh: int, rw: int, ph: int, pw: int, mul=2.0, add=-1.0, cpad=None):
    """DiffUIE.forward pre-processing (unifie.py:124-134) + x*2-1 + layout: fp32 NCHW -> bf16 NHWC [N, rh+ph, rw+pw, cpad]."""
    img = img.contiguous().float()
    n, c, h, w_ = img.shape
    cpad = cpad or round_up(c, 8)
    out = torch.empty((n, rh + ph, rw + pw, cpad), dtype=_act, device=img.device)
    check(lib.ur_image_resize_pad_nhwc(img.data_ptr(), out.data_ptr(), n, c, h, w_, rh, rw, ph, pw, cpad, mul, add, _dt(), _stream()))
    return out


def image_unpad_resize(x: torch.Tensor, c: int, crop_hw, out_hw, mul=1.0, add=0.0, quantize=False):
    """Post-processing (unifie.py:164-168 [+ eval_image_restoration.py:71 when quantize]): NHWC -> crop -> bicubic -> fp32 NCHW."""
    n, xh, xw, ld = x.shape
    out = torch.empty((n, c, out_hw[0], out_hw[1]), dtype=torch.float32, device=x.device)
    check(lib.ur_image_unpad_resize_nchw(x.data_ptr(), int(x.dtype == torch.float32), out.data_ptr(), n, c, xh, xw, ld, crop_hw[0],
                                         crop_hw[1], out_hw[0], out_hw[1], mul, add, int(quantize), _dt() if x.dtype == torch.float32 else _dt(x), _stream()))
    return out


def nhwc_to_nchw(x: torch.Tensor, c=None, mul=1.0, add=0.0):
    n, h, w_, ld = x.shape
    c = c or ld
    out = torch.empty((n, c, h, w_), dtype=torch.float32, device=x.device)
    check(lib.ur_nhwc_to_nchw_f32(x.data_ptr(), int(x.dtype == torch.float32), out.data_ptr(), n, c, h, w_, ld, mul, add,
                                  _dt() if x.dtype == torch.float32 else _dt(x), _stream()))
    return out


def vae_sample(moments_f32, noise_nchw, clat, scale):
    n, h, w_, ld = moments_f32.shape
    z = torch.empty((n, h, w_, 8), dtype=torch.float32, device=moments_f32.device)
    zb = torch.empty((n, h, w_, 8), dtype=_act, device=moments_f32.device)
    check(lib.ur_vae_sample(moments_f32.data_ptr(), ld, noise_nchw.data_ptr(), z.data_ptr(), zb.data_ptr(), n, h * w_, clat, 8,
                            scale, _dt(), _stream()))
    return z, zb


def add_noise(z0, noise_nchw, clat, sa, sb):
    n, h, w_, cp = z0.shape
    zt, zb = torch.empty_like(z0), torch.empty(z0.shape, dtype=_act, device=z0.device)
    check(lib.ur_add_noise(z0.data_ptr(), noise_nchw.data_ptr(), zt.data_ptr(), zb.data_ptr(), n, h * w_, clat, cp, sa, sb, _dt(), _stream()))
    return zt, zb


def ddim_step_(zt, zt_bf16, eps_f32, clat, c_x, c_e):
    cp = zt.shape[-1]
    check(lib.ur_ddim_step(zt.data_ptr(), eps_f32.data_ptr(), eps_f32.shape[-1], zt_bf16.data_ptr(), zt.numel() // cp, clat, cp,
                           c_x, c_e, _dt(zt_bf16), _stream()))


def f32_to_bf16(x, c, mul=1.0, cpad=8):
    ld = x.shape[-1]
    out = torch.empty((*x.shape[:-1], cpad), dtype=_act, device=x.device)
    check(lib.ur_f32_to_bf16_scaled(x.data_ptr(), ld, out.data_ptr(), x.numel() // ld, c, cpad, mul, _dt(), _stream()))
    return out


def profile_enable(on: bool):
    check(lib.ur_profile_enable(int(on)))


def profile_report() -> dict:
    import ctypes
    import json
    buf = ctypes.create_string_buffer(1 << 20)
    check(lib.ur_profile_report(buf, len(buf)))
    return json.loads(buf.value.decode())
