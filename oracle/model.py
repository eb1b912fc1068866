"""Oracle model graph: Controller, ControlledUNet, SkipConnectedAutoEncoder, DiffUIE (tests only).

Pure-torch fp32, NCHW, eager — numerically the reference's CPU path with the release defects of
SURVEY.md §3.5 skipped (FLOP probe + raise at unifie.py:43-53; TaskEditorV1c rename).
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import schedule
from .adapters import SPADE, CSCEAdapter, TaskFeatureAdapter, cfrm_blocks
from .blocks import (AutoencoderKL, DownBlock, MidBlock, ResnetBlock2D, TimestepEmbedding, UNet2DConditionModel,
                     gaussian_sample, sinusoidal_embedding)

# /root/reference/src/modules/diffuie/controller.py:29-45
stablesr_config = dict(in_channels=4, model_channels=256, out_channels=256, num_res_blocks=2, dropout=0,
                       channel_mult=(1, 1, 2, 2), downsample_type="conv", num_heads=4,
                       down_block_types=("AttnDownBlock2D",) * 3 + ("DownBlock2D",), mid_block_type="UNetMidBlock2D")


class Controller(nn.Module):
    """/root/reference/src/modules/diffuie/controller.py:65-220."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, dropout, channel_mult,
                 downsample_type, num_heads, down_block_types, mid_block_type, groups=32):
        super().__init__()
        if mid_block_type != "UNetMidBlock2D":
            raise NotImplementedError(mid_block_type)
        self.model_channels = model_channels
        temb = model_channels * 4
        self.time_embedding = TimestepEmbedding(model_channels, temb)
        self.conv_in = nn.Conv2d(in_channels, model_channels, 3, padding=1)
        self.down_blocks = nn.ModuleList()
        chans, out = [], model_channels
        for i, kind in enumerate(down_block_types):
            cin, out = out, model_channels * channel_mult[i]
            last = i == len(channel_mult) - 1
            self.down_blocks.append(DownBlock(cin, out, temb, attn="self" if kind == "AttnDownBlock2D" else None,
                                              head_dim=out // num_heads, add_downsample=not last,
                                              layers=num_res_blocks, groups=groups, eps=1e-5))
            chans.append(out)
        self.middle_block = MidBlock(out, temb, "self", head_dim=out // num_heads, groups=groups, eps=1e-5)
        self.fea_tran = nn.ModuleList([ResnetBlock2D(c, out_channels, temb, groups, 1e-5) for c in chans])
        for m in self.modules():                      # zero-conv init, controller.py:174-185
            if isinstance(m, ResnetBlock2D):
                nn.init.zeros_(m.conv2.weight), nn.init.zeros_(m.conv2.bias)
        for m in self.modules():
            if hasattr(m, "to_out") and hasattr(m, "group_norm"):
                nn.init.zeros_(m.to_out[0].weight), nn.init.zeros_(m.to_out[0].bias)

    def forward(self, x, timesteps, encoder_hidden_states=None):
        emb = self.time_embedding(sinusoidal_embedding(timesteps, self.model_channels))
        feats = []
        h = self.conv_in(x)
        for blk in self.down_blocks:
            h, states = blk(h, emb)
            feats.append(states[-2])
        feats[-1] = self.middle_block(h, emb)
        return {f.size(-1): self.fea_tran[i](f, emb) for i, f in enumerate(feats)}


class ControlledUNet(nn.Module):
    """/root/reference/src/modules/diffuie/base_model.py:13-245 (control_type 'scedit' or 'spade')."""

    def __init__(self, unet: UNet2DConditionModel, control_type: str, null_embeds: Optional[torch.Tensor] = None,
                 cond_channels: int = 256):
        super().__init__()
        self.unet = unet
        cross_dim = unet.down_blocks[0].attentions[0].transformer_blocks[0].attn2.to_k.in_features
        self.register_buffer("null_embeds", null_embeds if null_embeds is not None else torch.zeros(1, 77, cross_dim))
        self.control_type = control_type
        if control_type == "spade":                                  # base_model.py:32-37: a SPADE on every UNet resnet
            for m in list(unet.modules()):
                if isinstance(m, ResnetBlock2D):
                    m.spade = SPADE(m.conv2.out_channels, cond_channels)
        elif control_type == "scedit":
            chans = [unet.conv_in.out_channels]
            for blk in unet.down_blocks:
                chans += [r.conv2.out_channels for r in blk.resnets]
                if blk.downsamplers is not None:
                    chans.append(chans[-1])
            self.csc_editors = nn.ModuleList([CSCEAdapter(c, c, cond_channels) for c in chans])
        else:
            raise ValueError(f"control_type '{control_type}' not supported")

    def forward(self, sample, control, timesteps):
        u = self.unet
        ctx = self.null_embeds.expand(sample.shape[0], -1, -1)
        emb = u.time_embedding(sinusoidal_embedding(timesteps, u.time_proj_dim).to(sample.dtype))
        h = u.conv_in(sample)
        skips = [h]
        sp = control if self.control_type == "spade" else None        # spade_resnet (base_model.py:56-92) vs _resnet (:47-54)
        for blk in u.down_blocks:
            for i, res in enumerate(blk.resnets):
                h = res(h, emb, sp)
                if blk.attn_kind == "cross":
                    h = blk.attentions[i](h, ctx)
                skips.append(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0](h)
                skips.append(h)
        h = u.mid_block(h, emb, ctx, sp)
        if hasattr(self, "csc_editors"):                              # base_model.py:233-238
            skips = [ed(s, control[s.shape[-1]]) for ed, s in zip(self.csc_editors, skips)]
        for blk in u.up_blocks:
            for i, res in enumerate(blk.resnets):
                h = res(torch.cat([h, skips.pop()], dim=1), emb, sp)
                if blk.attn_kind == "cross":
                    h = blk.attentions[i](h, ctx)
            if blk.upsamplers is not None:
                h = blk.upsamplers[0](h)
        return u.conv_out(F.silu(u.conv_norm_out(h)))


class SkipConnectedAutoEncoder(nn.Module):
    """/root/reference/src/modules/diffuie/autoencoder.py:74-184 with the patched forwards :11-72."""

    def __init__(self, vae: AutoencoderKL, fr_type: Optional[str] = None, tedit: Optional[dict] = None,
                 fr_depths=(1, 1, 9)):
        super().__init__()
        self.vae = vae
        enc_ch = [b.resnets[-1].conv2.out_channels for b in vae.encoder.down_blocks]
        if fr_type == "CFRM":
            vae.encoder.fr_blocks = cfrm_blocks(enc_ch[:3], fr_depths)
        elif fr_type is not None:
            raise ValueError("Invalid fr_type")
        self.task_list = []
        if tedit:
            if tedit["type"] != "TFA":
                raise KeyError("%s is not defined in the taskeditor!, please select ['TFA']" % tedit["type"])
            self.task_list = list(tedit["task"])
            pl, top = tedit["prompt_len"], enc_ch[-1]
            vae.decoder.task_prompts = nn.ParameterDict({t: nn.Parameter(torch.zeros(pl, enc_ch[2])) for t in self.task_list})
            vae.decoder.task_editors = nn.ModuleList([
                TaskFeatureAdapter(top, enc_ch[2], prompt_len=pl),
                TaskFeatureAdapter(top, enc_ch[1], prompt_len=pl),
                TaskFeatureAdapter(top, enc_ch[0], prompt_len=pl, last_layer=True)])

    def encode(self, images, enable_fr=False, noise=None):
        enc = self.vae.encoder
        h = enc.conv_in(images * 2 - 1)
        res = []
        for i, blk in enumerate(enc.down_blocks[:-1]):
            h = blk(h)
            if enable_fr:
                h = enc.fr_blocks[i](h)
            res.append(h)
        h = enc.mid_block(enc.down_blocks[-1](h))
        moments = self.vae.quant_conv(enc.conv_out(F.silu(enc.conv_norm_out(h))))
        if noise is None:
            noise = torch.randn_like(moments[:, : moments.shape[1] // 2])
        return gaussian_sample(moments, noise) * self.vae.scaling_factor, res

    def decode(self, latents, res_samples, task):
        dec = self.vae.decoder
        h = dec.mid_block(dec.conv_in(self.vae.post_quant_conv(latents / self.vae.scaling_factor)))
        if not hasattr(dec, "task_editors"):          # tedit=None: the decoder forward is not patched (autoencoder.py:107-110)
            for blk in dec.up_blocks:
                h = blk(h)
            return (dec.conv_out(F.silu(dec.conv_norm_out(h))) + 1) / 2
        cond = dec.task_prompts[task].unsqueeze(0).expand(latents.shape[0], -1, -1)
        for i, blk in enumerate(dec.up_blocks[:-1]):
            h, cond = dec.task_editors[i](h, res_samples[-i - 1], cond)
            h = blk(h)
        h = dec.up_blocks[-1](h)
        return (dec.conv_out(F.silu(dec.conv_norm_out(h))) + 1) / 2


def resize_pad_plan(h: int, w: int):
    """Integer shape arithmetic of unifie.py:121-134 -> (resized_h, resized_w, pad_h, pad_w)."""
    if h < 512 or w < 512:
        s = 512 / min(h, w)
        h, w = round(h * s), round(w * s)
    return h, w, (64 - h % 64) % 64, (64 - w % 64) % 64


def center_crop_box(h: int, w: int, upper=512):
    """crop_tensor, /root/reference/src/core/base/eval_image_restoration.py:113-136."""
    ch, cw = min(h, upper), min(w, upper)
    return h // 2 - ch // 2, h // 2 + ch // 2, w // 2 - cw // 2, w // 2 + cw // 2


class DiffUIE(nn.Module):
    """/root/reference/src/modules/diffuie/unifie.py:22-169."""

    def __init__(self, frenc=None, cnet=None, tedit=None, *, unet_cfg=None, vae_cfg=None, controller_cfg=None,
                 null_embeds=None, fr_depths=(1, 1, 9)):
        super().__init__()
        self.fr_type = frenc["type"] if frenc else None
        self.control_type = cnet["type"] if cnet else None
        self.tedit = tedit if tedit else None
        self.ae = SkipConnectedAutoEncoder(AutoencoderKL(**(vae_cfg or {})), self.fr_type, self.tedit, fr_depths)
        if self.control_type:
            self.controller = Controller(**(controller_cfg or stablesr_config))
            self.base_model = ControlledUNet(UNet2DConditionModel(**(unet_cfg or {})), self.control_type, null_embeds,
                                             (controller_cfg or stablesr_config)["out_channels"])
            self.register_buffer("train_timesteps", torch.tensor([249, 499, 749, 999, 999, 999], dtype=torch.int64))
            self.num_inference_steps = cnet["num_inference_steps"]
            self.timesteps = schedule.ddim_timesteps(self.num_inference_steps)

    def diffuse(self, latents, timesteps=None, noise=None):
        if timesteps is None:
            timesteps = self.train_timesteps[torch.randint(0, len(self.train_timesteps), (latents.size(0),))]
        noise = torch.randn_like(latents) if noise is None else noise
        return schedule.add_noise(latents, noise, timesteps), noise, timesteps

    def predict_z0(self, latents, conditions, timesteps):
        eps = self.base_model(latents, self.controller(conditions, timesteps), timesteps)
        a = schedule.alphas_cumprod()[timesteps].view(-1, 1, 1, 1)
        return (latents - (1 - a) ** 0.5 * eps) / a ** 0.5

    @torch.no_grad()
    def forward(self, images, task, noise=None, return_latents=False):
        """noise = (eps_vae, eps_t999) makes the two RNG draws (autoencoder.py:152, unifie.py:87) explicit."""
        org_h, org_w = images.shape[-2:]
        h, w, pad_h, pad_w = resize_pad_plan(org_h, org_w)
        if (h, w) != (org_h, org_w):
            images = F.interpolate(images, (h, w), mode="bicubic", align_corners=False, antialias=False)
        if pad_h or pad_w:
            images = F.pad(images, (0, pad_w, 0, pad_h), mode="reflect")
        n_vae, n_t = noise if noise is not None else (None, None)
        z0, mids = self.ae.encode(images, enable_fr=self.fr_type is not None, noise=n_vae)
        zt = z0
        if self.control_type:
            t999 = torch.full((len(images),), 999, dtype=torch.int64)
            zt, _, _ = self.diffuse(z0, t999, n_t)
            for t in self.timesteps:
                ts = torch.tensor([int(t)], dtype=torch.int64)
                eps = self.base_model(zt, self.controller(z0, ts), ts)
                zt = schedule.ddim_step(eps, int(t), zt, self.num_inference_steps)
                if getattr(self, "trace_zt", None) is not None:
                    self.trace_zt.append(zt.clone())                          # per-step latents for the trajectory parity test
        preds = self.ae.decode(zt, mids, task)[..., :h, :w]
        preds = F.interpolate(preds, (org_h, org_w), mode="bicubic", align_corners=False, antialias=False)
        return (preds, z0, zt) if return_latents else preds
