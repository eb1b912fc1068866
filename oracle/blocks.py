"""Pure-torch fp32 restatement of the un-vendored diffusers (0.29.0) blocks the hot path uses.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference reaches these through
  /root/reference/src/modules/diffuie/unifie.py:40,60        (AutoencoderKL / UNet2DConditionModel)
  /root/reference/src/modules/diffuie/controller.py:3-11,86-170 (Timesteps, TimestepEmbedding,
      get_down_block, UNetMidBlock2D, ResnetBlock2D)
  /root/reference/src/modules/diffuie/base_model.py:94-209     (manual walk over the HF UNet tree)
  /root/reference/src/modules/diffuie/autoencoder.py:11-72     (patched encoder/decoder forwards)
Parameter names follow the HF checkpoints.  "parity unpinned" beyond
structure: diffusers is not installed here; names and parameter counts are verified in tests.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- embeddings
def sinusoidal_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): output is [cos, sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    ang = t.reshape(-1, 1).float() * freqs.reshape(1, -1)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, time_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, time_dim)
        self.linear_2 = nn.Linear(time_dim, time_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


# ----------------------------------------------------------------------------- resnet / resample
class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels=None, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, cout) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None, control=None):
        """control: {width: feature map}; used only when a `spade` module was grafted on (base_model.py:56-92)."""
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if control is not None and hasattr(self, "spade"):
            h = self.spade(h, control[h.shape[-1]])          # base_model.py:85-86
        sc = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return sc + h


class Downsample2D(nn.Module):
    def __init__(self, c, padding):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:                      # VAE encoder: pad right/bottom by one
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


# ----------------------------------------------------------------------------- attention
def _sdpa(q, k, v, heads):
    b, tq, c = q.shape
    d = c // heads
    q = q.view(b, tq, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    if tq * k.shape[2] > (1 << 22):
        # long sequences (>= 2048 x 2048; 16384 tokens at 1024x1024): same mathematics through PyTorch's fused fp32 kernel, which
        # does not materialise the [heads, Tq, Tk] score tensor (5 GB per attention at 16384 tokens) - minutes -> seconds on CPU
        return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, tq, c)
    w = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)
    return (w @ v).transpose(1, 2).reshape(b, tq, c)


class _ToOut(nn.ModuleList):
    """`to_out.0` = Linear, `to_out.1` = Dropout (no params)."""

    def __init__(self, c):
        super().__init__([nn.Linear(c, c), nn.Identity()])


class AttentionBlock(nn.Module):
    """Legacy spatial self-attention block (VAE mid block, Controller AttnDownBlock2D / mid)."""

    def __init__(self, c, head_dim, groups=32, eps=1e-5):
        super().__init__()
        self.heads = c // head_dim
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = _ToOut(c)

    def forward(self, x):
        b, c, hh, ww = x.shape
        h = self.group_norm(x).view(b, c, hh * ww).transpose(1, 2)
        o = _sdpa(self.to_q(h), self.to_k(h), self.to_v(h), self.heads)
        o = self.to_out[0](o)
        return o.transpose(1, 2).reshape(b, c, hh, ww) + x


class CrossAttention(nn.Module):
    def __init__(self, c, heads, kv_dim=None):
        super().__init__()
        self.heads = heads
        kv_dim = kv_dim or c
        self.to_q = nn.Linear(c, c, bias=False)
        self.to_k = nn.Linear(kv_dim, c, bias=False)
        self.to_v = nn.Linear(kv_dim, c, bias=False)
        self.to_out = _ToOut(c)

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        return self.to_out[0](_sdpa(self.to_q(x), self.to_k(ctx), self.to_v(ctx), self.heads))


class GEGLU(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.proj = nn.Linear(c, inner * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(c, 4 * c), nn.Identity(), nn.Linear(4 * c, c)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, c, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(c)
        self.attn1 = CrossAttention(c, heads)
        self.norm2 = nn.LayerNorm(c)
        self.attn2 = CrossAttention(c, heads, cross_dim)
        self.norm3 = nn.LayerNorm(c)
        self.ff = FeedForward(c)

    def forward(self, h, ctx):
        h = h + self.attn1(self.norm1(h))
        h = h + self.attn2(self.norm2(h), ctx)
        return h + self.ff(self.norm3(h))


class Transformer2DModel(nn.Module):
    """use_linear_projection=True, one layer."""

    def __init__(self, c, heads, cross_dim, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.proj_in = nn.Linear(c, c)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(c, heads, cross_dim)])
        self.proj_out = nn.Linear(c, c)

    def forward(self, x, ctx):
        b, c, hh, ww = x.shape
        h = self.norm(x).permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = self.proj_in(h)
        h = self.transformer_blocks[0](h, ctx)
        h = self.proj_out(h)
        return h.reshape(b, hh, ww, c).permute(0, 3, 1, 2) + x


# ----------------------------------------------------------------------------- UNet blocks
class DownBlock(nn.Module):
    """DownBlock2D / CrossAttnDownBlock2D / AttnDownBlock2D (attn: None | 'cross' | 'self')."""

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

    def forward(self, h, temb, ctx=None):
        states = []
        for i, res in enumerate(self.resnets):
            h = res(h, temb)
            if self.attn_kind == "cross":
                h = self.attentions[i](h, ctx)
            elif self.attn_kind == "self":
                h = self.attentions[i](h)
            states.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            states.append(h)
        return h, states


class MidBlock(nn.Module):
    """UNetMidBlock2DCrossAttn (attn='cross') / UNetMidBlock2D (attn='self')."""

    def __init__(self, c, temb, attn, heads=None, head_dim=None, cross_dim=None, groups=32, eps=1e-5):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attn_kind = attn
        if attn == "cross":
            self.attentions = nn.ModuleList([Transformer2DModel(c, heads, cross_dim, groups)])
        else:
            self.attentions = nn.ModuleList([AttentionBlock(c, head_dim, groups, eps)])

    def forward(self, h, temb=None, ctx=None, control=None):
        h = self.resnets[0](h, temb, control)
        h = self.attentions[0](h, ctx) if self.attn_kind == "cross" else self.attentions[0](h)
        return self.resnets[1](h, temb, control)


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


UNET_SD21 = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 heads=(5, 10, 20, 20), cross_dim=1024, groups=32, layers=2)


class UNet2DConditionModel(nn.Module):
    """SD-2.1 layout: 3 CrossAttnDown + DownBlock2D, cross-attn mid, UpBlock2D + 3 CrossAttnUp."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 heads=(5, 10, 20, 20), cross_dim=1024, groups=32, layers=2):
        super().__init__()
        ch = list(block_out_channels)
        self.time_proj_dim = ch[0]
        temb = ch[0] * 4
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
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
        self.conv_norm_out = nn.GroupNorm(groups, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)


# ----------------------------------------------------------------------------- VAE
class EncDownBlock(nn.Module):
    def __init__(self, cin, cout, add_downsample, groups, layers=2):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, 0)]) if add_downsample else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
        return h


class DecUpBlock(nn.Module):
    def __init__(self, cin, cout, add_upsample, groups, layers=3):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, ch, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i in range(len(ch)):
            cin, out = out, ch[i]
            self.down_blocks.append(EncDownBlock(cin, out, i < len(ch) - 1, groups))
        self.mid_block = MidBlock(ch[-1], None, "self", head_dim=ch[-1], groups=groups, eps=1e-6)
        self.conv_norm_out = nn.GroupNorm(groups, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * latent_channels, 3, padding=1)


class Decoder(nn.Module):
    def __init__(self, out_channels, latent_channels, ch, groups):
        super().__init__()
        rev = list(ch)[::-1]
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], None, "self", head_dim=rev[0], groups=groups, eps=1e-6)
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i in range(len(rev)):
            prev, out = out, rev[i]
            self.up_blocks.append(DecUpBlock(prev, out, i < len(rev) - 1, groups))
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)


VAE_SD = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), groups=32)


class AutoencoderKL(nn.Module):
    scaling_factor = 0.18215

    def __init__(self, in_channels=3, out_channels=3, latent_channels=4,
                 block_out_channels=(128, 256, 512, 512), groups=32):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, groups)
        self.decoder = Decoder(out_channels, latent_channels, block_out_channels, groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)


def gaussian_sample(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution(moments).sample() with the noise passed in."""
    mean, logvar = moments.chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise
