"""UniRestore's adapters on the HIP ops: CSCEAdapter (SC-Tuner), NAFBlock / AdaNAFV2 (CFRM), TaskFeatureAdapter (TFA).

Same constructor signatures, parameter names and forward semantics as the reference classes
(scedit.py:24-38, nafnet_arch.py:28-131, cfrm.py:12-54, taskeditor.py:10-108); `forward` takes/returns NCHW fp32
tensors like the reference operators, `run` is the NHWC bf16 fast path the model graph uses.
"""
import os

import torch
import torch.nn as nn

from .. import chain, ops
from ..ops import UR_ACT_GATE, UR_ACT_GELU, UR_ACT_NONE, UR_ACT_RELU, UR_ACT_TANH
from . import nn as _nn
from .nn import DEV, Conv2d, LayerNorm, Linear, GroupNorm


# LayerNorm2d of the NAFBlocks folded into the consuming 1x1 convs (the mechanism of the transformer blocks' LayerNorms): 25 of the
# 28 LayerNorm passes of a forward disappear.  UR_FUSE_LN2D=0: separate passes (A/B).
FUSE_LN2D = os.environ.get("UR_FUSE_LN2D", "1") == "1"


def _named(**mods):
    return nn.ModuleDict({k.lstrip("_"): v for k, v in mods.items()})


def _finite16(t):
    """Operator-level fp16 entry points check what they return (fp16 conversions overflow to inf: csrc/common.h Act<true>::pack2)."""
    if ops.act_dtype() == torch.float16 and not bool(torch.isfinite(t).all()):
        raise FloatingPointError("fp16 activation overflow (|x| > 65504): the result is not finite.  Use dtype='bf16' for these weights")
    return t


def _to_nhwc(x):
    return ops.nchw_to_nhwc(x.to(DEV))


def _fresh():
    """Operator-level entry (reference signature).  (Statistics live in per-tensor planes: nothing to reset.)"""


class CSCEAdapter(nn.Module):
    """out = tuner(x + proj(cond)) + proj(cond) + x, as 3 GEMMs with the adds in their epilogues."""

    def __init__(self, c_in, c_emb, c_cond):
        super().__init__()
        self.proj = Conv2d(c_cond, c_in, 1)
        self.tuner = _named(_0=Conv2d(c_in, c_emb, 1), _2=Conv2d(c_emb, c_in, 1))

    def run(self, x, condition):
        n, hh, ww, c = x.shape
        if (_nn.CHAIN and c == chain.CHAIN_C and condition.shape[-1] == 256 and self.tuner["0"].out_channels == c and
                (hh * ww) % chain.CHAIN_TOK == 0 and x.is_contiguous() and condition.is_contiguous() and
                tuple(condition.shape[:3]) == (n, hh, ww) and condition.dtype == x.dtype):
            key = ("cache", "chain", ops.act_dtype())          # one launch: the token-stationary chain (csrc/tchain.hip, kind CSCE)
            if key not in self.__dict__:
                self.__dict__[key] = chain.pack_csce(self.proj.weight, self.proj.bias, self.tuner["0"].weight, self.tuner["0"].bias,
                                                     self.tuner["2"].weight, self.tuner["2"].bias, DEV)
            return chain.csce_fused(x, condition, self.__dict__[key])
        if tuple(condition.shape[:3]) != (n, hh, ww):
            raise ValueError(f"CSCEAdapter: condition {tuple(condition.shape)} does not cover x {tuple(x.shape)}")
        s = ops.conv(condition, self.proj.packed(), residual=x)               # s = x + proj(cond)
        h = ops.conv(s, self.tuner["0"].packed(), act=UR_ACT_GELU)
        return ops.conv(h, self.tuner["2"].packed(), residual=s, gn=True)     # tuner(s) + s (feeds a GroupNorm in the up path)

    def forward(self, x, condition):
        _fresh()
        return _finite16(ops.nhwc_to_nchw(self.run(_to_nhwc(x), _to_nhwc(condition)), c=x.shape[1]))


class SPADE(nn.Module):
    """Alternative control path (spade.py:29-71): out = GroupNorm(x) * (1 + gamma(seg)) + beta(seg), gamma / beta from a
    shared conv3x3 + ReLU.  gamma and beta are ONE conv (weights concatenated); the modulation and the resnet's residual
    add are one elementwise kernel."""

    def __init__(self, norm_nc, label_nc=128, config_text="spadegroup3x3", nhidden=128):
        super().__init__()
        if config_text != "spadegroup3x3":
            raise NotImplementedError(config_text)
        self.param_free_norm = GroupNorm(32, norm_nc)          # affine despite the name (spade.py:40)
        self.mlp_shared = _named(_0=Conv2d(label_nc, nhidden, 3, padding=1))
        self.mlp_gamma = Conv2d(nhidden, norm_nc, 3, padding=1)
        self.mlp_beta = Conv2d(nhidden, norm_nc, 3, padding=1)

    def _gb(self):
        key = ("gb", ops.act_dtype())
        if key not in self.__dict__:
            w = torch.cat([self.mlp_gamma.weight.detach().float(), self.mlp_beta.weight.detach().float()], 0)
            b = torch.cat([self.mlp_gamma.bias.detach().float(), self.mlp_beta.bias.detach().float()], 0)
            self.__dict__[key] = ops.pack_conv(w, b, DEV)
        return self.__dict__[key]

    def run(self, x, seg, residual=None):
        """x [B,h,w,C] bf16 (with or without producer sums), seg [B,h2,w2,label_nc] -> modulated x (+ residual)."""
        if seg.shape[1:3] != x.shape[1:3]:                      # F.interpolate(mode="nearest"): src = floor(dst * in / out)
            iy = (torch.arange(x.shape[1], device=seg.device) * seg.shape[1]) // x.shape[1]
            ix = (torch.arange(x.shape[2], device=seg.device) * seg.shape[2]) // x.shape[2]
            seg = seg[:, iy][:, :, ix].contiguous()
        actv = ops.conv(seg, self.mlp_shared["0"].packed(), act=UR_ACT_RELU)
        gb = ops.conv(actv, self._gb())
        return ops.spade_modulate(self.param_free_norm.run(x), gb, residual)

    def forward(self, x, segmap):
        _fresh()
        return _finite16(ops.nhwc_to_nchw(self.run(_to_nhwc(x), _to_nhwc(segmap)), c=x.shape[1]))


class LayerNorm2d(LayerNorm):
    """LayerNorm over C of an image tensor; in NHWC that is a plain row LayerNorm (eps 1e-6, timm)."""

    def __init__(self, c):
        super().__init__(c, eps=1e-6)


class NAFBlock(nn.Module):
    def __init__(self, c, DW_Expand=2, FFN_Expand=2, drop_out_rate=0.0):
        super().__init__()
        assert DW_Expand == 2 and FFN_Expand == 2 and drop_out_rate == 0.0
        self.conv1 = Conv2d(c, 2 * c, 1)
        self.conv2 = Conv2d(2 * c, 2 * c, 3, padding=1, groups=2 * c)
        self.conv3 = Conv2d(c, c, 1)
        self.sca = _named(_1=Conv2d(c, c, 1))
        self.conv4 = Conv2d(c, 2 * c, 1)
        self.conv5 = Conv2d(c, c, 1)
        self.norm1, self.norm2 = LayerNorm2d(c), LayerNorm2d(c)
        self.beta = nn.Parameter(torch.zeros(1, c, 1, 1))
        self.gamma = nn.Parameter(torch.zeros(1, c, 1, 1))

    def _dw(self):
        if "dw" not in self.__dict__:
            c2 = self.conv2.weight.shape[0]
            self.__dict__["dw"] = (self.conv2.weight.detach().float().view(c2, 9).t().contiguous().to(DEV),
                                   self.conv2.bias.detach().float().to(DEV),
                                   self.sca["1"].weight.detach().float().view(c2 // 2, c2 // 2).contiguous().to(DEV),
                                   self.sca["1"].bias.detach().float().to(DEV))
        return self.__dict__["dw"]

    def _ln_folded(self, conv, norm, pair):
        """1x1 conv of LayerNorm2d(x): LayerNorm over C folded into the GEMM (w' = W * gamma, b' = W.beta + b; the epilogue applies
        rstd * (acc - mean * colsum) from the row sums the PRODUCER of x left) - no LayerNorm pass, no normalised tensor."""
        key = ("cache", "ln", id(conv), pair, ops.act_dtype())
        if key not in self.__dict__:
            self.__dict__[key] = ops.pack_linear_ln(conv.weight.detach().float().flatten(1), conv.bias, norm.weight, norm.bias, norm.eps, DEV, pair=pair)
        return self.__dict__[key]

    def run(self, inp):
        w9c, b2, wsca, bsca = self._dw()
        st = ops.ln_of(inp) if FUSE_LN2D else None
        if st is not None:                        # the producer (previous NAFBlock / AdaNAFV2.pwconv) left per-pixel channel sums
            x = ops.conv(inp, self._ln_folded(self.conv1, self.norm1, False), ln_stats=st)
        else:
            x = ops.conv(self.norm1.run(inp), self.conv1.packed())
        x = ops.dwconv3x3(x, w9c, b2, gate=True)                               # depthwise + SimpleGate
        s = ops.linear_f32(ops.avgpool(x), wsca, bsca)                         # simplified channel attention
        x = ops.scale_channels(x, s)
        y = ops.conv(x, self.conv3.packed(scale=self.beta), residual=inp, rows=FUSE_LN2D)      # inp + conv3(x)*beta (beta folded)
        if FUSE_LN2D:
            x = ops.conv(y, self._ln_folded(self.conv4, self.norm2, True), ln_stats=ops.ln_of(y), act=UR_ACT_GATE)
        else:
            x = ops.conv(self.norm2.run(y), self.conv4.packed(pair=True), act=UR_ACT_GATE)
        return ops.conv(x, self.conv5.packed(scale=self.gamma), residual=y, gn=True, rows=FUSE_LN2D)    # y + conv5(x)*gamma

    def forward(self, inp):
        _fresh()
        return _finite16(ops.nhwc_to_nchw(self.run(_to_nhwc(inp)), c=inp.shape[1]))


class AdaNAFV2(nn.Module):
    GROUPS = 16

    def __init__(self, c, DW_Expand=2, FFN_Expand=2, drop_out_rate=0.0):
        super().__init__()
        g, w = self.GROUPS, 4 * c
        self.conv_in = Conv2d(c, w, 1)
        self.group_norm = GroupNorm(g, w)
        self.group_conv = Conv2d(w, w, 3, padding=1, groups=g)
        self.intra_group_attn = _named(_1=Conv2d(w, w, 1, groups=g))
        self.inter_group_attn = _named(_1=Conv2d(w, g, 1))
        self.pwconv = Conv2d(w, c, 1)
        self.nafblock = NAFBlock(c, DW_Expand, FFN_Expand, drop_out_rate)

    def _vecs(self):
        if "vecs" not in self.__dict__:
            ia, ie = self.intra_group_attn["1"], self.inter_group_attn["1"]
            self.__dict__["vecs"] = (ia.weight.detach().float().flatten(1).contiguous().to(DEV), ia.bias.detach().float().to(DEV),
                                     ie.weight.detach().float().flatten(1).contiguous().to(DEV), ie.bias.detach().float().to(DEV))
        return self.__dict__["vecs"]

    def _group_conv_packed(self):
        """The grouped 3x3 conv (16 groups of 4c/16 channels: cfrm.py:20-21).  Groups narrower than a 64-channel halo chunk would
        take the generic batched kernel (a 64-byte gather per pixel and tap: 2.19 ms at 256 x 256 x 512); they are densified instead
        into block-diagonal 128 -> 128 convolutions (zeros outside a group's own block: same sums, 2-4x the MFMA work, all of it
        at the halo kernel's rate) and run as 4c/128 "groups" of 128 channels.  Only for the measured widths (32- and 64-channel
        groups = the production CFRM): narrower groups would pay 8-16x the MFMA work and weight bytes and stay on the batched kernel."""
        key = ("pk", "gc", ops.act_dtype())
        if key not in self.__dict__:
            gc = self.group_conv
            w, cg, g = gc.weight.detach().float(), gc.weight.shape[1], gc.groups          # [W, Cg, 3, 3]
            if 32 <= cg < 128 and w.shape[0] % 128 == 0 and 128 % cg == 0 and os.environ.get("UR_GC_DENSE", "1") == "1":
                wd = torch.zeros(w.shape[0], 128, 3, 3)
                o = torch.arange(w.shape[0])
                off = ((o // cg) * cg) % 128                                                 # first input channel of o's group inside its 128-block
                for j in range(cg):
                    wd[o, off + j] = w[:, j].cpu()
                self.__dict__[key] = ops.pack_conv(wd, gc.bias, DEV, groups=w.shape[0] // 128, group_halo=True)
            else:
                self.__dict__[key] = ops.pack_conv(w, gc.bias, DEV, groups=g)      # (128-channel groups at 64 x 64: 16 half-round halo launches measured slower, 498 vs 324 us)
        return self.__dict__[key]

    def run(self, inp):
        g = self.GROUPS
        wia, bia, wie, bie = self._vecs()
        x = self.group_norm.run(ops.conv(inp, self.conv_in.packed(), gn=True))
        x = ops.conv(x, self._group_conv_packed(), act=UR_ACT_GELU)            # grouped 3x3 (+GELU)
        pooled = ops.avgpool(x)
        s_intra = ops.linear_f32(pooled, wia, bia, groups=g)                   # per-channel scale
        pooled2 = ops.vec_mul_group(pooled, s_intra, s_intra.shape[1])         # mean(x*s) = s*mean(x)
        iga = ops.linear_f32(pooled2, wie, bie)                                # per-group scale
        x = ops.scale_channels(x, ops.vec_mul_group(s_intra, iga, g))
        return self.nafblock.run(ops.conv(x, self.pwconv.packed(), residual=inp, rows=FUSE_LN2D))    # (row sums for the NAFBlock's norm1)

    def forward(self, inp):
        _fresh()
        return _finite16(ops.nhwc_to_nchw(self.run(_to_nhwc(inp)), c=inp.shape[1]))


class _Seq(nn.Sequential):
    def run(self, x):
        for m in self:
            x = m.run(x)
        return x

    def forward(self, x):
        _fresh()
        return _finite16(ops.nhwc_to_nchw(self.run(_to_nhwc(x)), c=x.shape[1]))


def cfrm_blocks(channels=(128, 256, 512), depths=(1, 1, 9)) -> nn.ModuleList:
    """fr_blocks wiring (autoencoder.py:91-98)."""
    return nn.ModuleList([_Seq(*[NAFBlock(c) for _ in range(n)], AdaNAFV2(c)) for c, n in zip(channels, depths)])


class TaskFeatureAdapter(nn.Module):
    def __init__(self, c_out=512, c_skip=256, prompt_len=1, last_layer=False):
        super().__init__()
        d = c_skip
        self.prompt_dim, self.prompt_len, self.last_layer = d, prompt_len, last_layer
        self.t_gate1 = Conv2d(c_skip, d, 1)
        self.t_gate2 = Conv2d(d, c_skip, 1)
        self.conv_out = Conv2d(c_skip + c_out, c_out, 1)

        def gate():
            return _named(_1=Conv2d(c_skip, c_skip, 3, padding=1), _3=Conv2d(c_skip, d * prompt_len, 3, padding=1))

        self.filter_gate, self.info_gate, self.content_trans = gate(), gate(), gate()
        self.out_gate = _named(_0=Linear(d * prompt_len, d))
        if not last_layer:
            self.prompt_trans = _named(_0=Linear(d, d // 2))

    def _fused(self):
        """The three gate branches share their input: stage 1 = one conv with 3x the output channels,
        stage 2 = one grouped conv (3 groups) whose epilogue is the global average pool."""
        key = ("fused", ops.act_dtype())
        if key not in self.__dict__:
            br = (self.filter_gate, self.info_gate, self.content_trans)
            w1 = torch.cat([b["1"].weight.detach().float() for b in br], 0)
            b1 = torch.cat([b["1"].bias.detach().float() for b in br], 0)
            w2 = torch.cat([b["3"].weight.detach().float() for b in br], 0)
            b2 = torch.cat([b["3"].bias.detach().float() for b in br], 0)
            self.__dict__[key] = (ops.pack_conv(w1, b1, DEV), ops.pack_conv(w2, b2, DEV, groups=3))
        return self.__dict__[key]

    def run(self, x, skip, condition):
        """x [B,h,w,c_out], skip [B,h,w,c_skip] bf16 NHWC; condition fp32 [B,T,D] -> (x', cond' or None)."""
        b, hh, ww, cs = skip.shape
        pc1, pc2 = self._fused()
        sn = ops.group_norm(skip, None, None, cs, 1e-5)                        # InstanceNorm2d (no affine); reuses fused sums
        h3 = ops.conv(sn, pc1, act=UR_ACT_GELU)
        # conv + AdaptiveAvgPool2d(1): where the conv's epilogue can leave the per-tile channel sums itself nothing but those
        # partial sums is written (no [B,h,w,3D] tensor); the finalize kernel adds them in a fixed order
        if ops.conv_plan(h3, pc2, gn=True, store=False).gn_fused:
            part, nparts = ops.conv(h3, pc2, gn=True, store=False)
            pooled = ops.gn_finalize_planes(part, nparts, hh * ww)
        else:
            pooled = ops.avgpool(ops.conv(h3, pc2, gn=True))
        upd = ops.tfa_prompt_update(pooled, condition.contiguous())
        wo, bo = self.out_gate["0"].dev_f32()
        o = ops.linear_f32(upd.view(b, -1), wo, bo, UR_ACT_TANH)               # [B, D]
        hs = ops.scale_channels(ops.conv(skip, self.t_gate1.packed()), o)
        skip2 = ops.conv(hs, self.t_gate2.packed(), residual=skip)
        x = ops.conv(x, self.conv_out.packed(), x2=skip2, residual=x, gn=True)  # x + conv_out(cat[x, skip])
        new_cond = None
        if not self.last_layer:
            wp, bp = self.prompt_trans["0"].dev_f32()
            new_cond = ops.linear_f32(upd.view(b * self.prompt_len, -1), wp, bp, UR_ACT_GELU).view(b, self.prompt_len, -1)
        return x, new_cond

    def forward(self, x, skip, condition):
        _fresh()
        y, c = self.run(_to_nhwc(x), _to_nhwc(skip), condition.to(DEV).float())
        return _finite16(ops.nhwc_to_nchw(y, c=x.shape[1])), c


TaskEditorV1c = TaskFeatureAdapter     # the reference imports it under this stale name (autoencoder.py:112)
