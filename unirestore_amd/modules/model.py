"""Model graph of the hot path on the HIP ops: Controller, ControlledUNet, SkipConnectedAutoEncoder, DiffUIE.

Mirrors the reference operator interface (names, constructor arguments, attribute paths, error behaviour):
  DiffUIE                    /root/reference/src/modules/diffuie/unifie.py:22-169
  SkipConnectedAutoEncoder   /root/reference/src/modules/diffuie/autoencoder.py:74-184 (+ patched forwards :11-72)
  Controller                 /root/reference/src/modules/diffuie/controller.py:65-220
  ControlledUNet             /root/reference/src/modules/diffuie/base_model.py:13-245
MI355X-first differences (results unchanged): NHWC bf16 activations, per-schedule time-embedding tables folded
into conv biases, constant cross-attention K/V computed once, the whole forward replayed as one hipGraph.
"""
import os
import weakref
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, schedule
from .adapters import SPADE, CSCEAdapter, TaskFeatureAdapter, cfrm_blocks
from . import nn as nnmod
from .nn import (DEV, AutoencoderKL, Conv2d, DownBlock, MidBlock, ResnetBlock2D, TimestepEmbedding, UNet2DConditionModel,
                 invalidate_packed, sinusoid_table)

stablesr_config = dict(in_channels=4, model_channels=256, out_channels=256, num_res_blocks=2, dropout=0,
                       channel_mult=(1, 1, 2, 2), downsample_type="conv", num_heads=4,
                       down_block_types=("AttnDownBlock2D",) * 3 + ("DownBlock2D",), mid_block_type="UNetMidBlock2D")


def _use_dtype(module):
    """Re-assert the 16-bit compute type of the DiffUIE that owns `module` (the op front end keeps it in a process global that
    constructing / calling another model may have changed).  Stand-alone sub-modules keep whatever type is current."""
    owner = module.__dict__.get("_ur_owner")
    owner = owner() if owner is not None else None
    if owner is not None and owner.__dict__.get("_stale"):       # some load_state_dict ran anywhere in the owner's tree
        owner.refresh()
    dt = module.__dict__.get("_ur_dtype")
    if dt is not None:
        ops.set_dtype(dt)


def _check_fp16(*tensors):
    """fp16 conversions overflow to inf (csrc/common.h Act<true>::pack2): every public operator-level entry point checks what it
    returns, so an fp16 range problem raises instead of handing the caller inf / NaN tensors (bf16 has fp32's range: no check)."""
    if ops.act_dtype() != torch.float16:
        return
    for t in tensors:
        for v in (t.values() if isinstance(t, dict) else (t,)):
            if not bool(torch.isfinite(v).all()):
                raise FloatingPointError("fp16 activation overflow (|x| > 65504): the result is not finite.  Use dtype='bf16' for these weights")


def _per_sample_timesteps(timesteps, batch):
    """Reference operator contract (controller.py:193-194, base_model.py:211-216): `timesteps` is a scalar / (1,) tensor shared by
    the batch, or a (B,) tensor with one timestep per sample.  Returns [t] or the B per-sample values (all-equal collapses to [t])."""
    ts = [int(t) for t in torch.as_tensor(timesteps).reshape(-1).tolist()]
    if len(ts) != 1 and len(ts) != batch:
        raise ValueError(f"timesteps must hold 1 or {batch} values, got {len(ts)}")
    return ts[:1] if len(set(ts)) == 1 else ts


def _resnets(module):
    return [m for m in module.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]


class Controller(nn.Module):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, dropout, channel_mult,
                 downsample_type, num_heads, down_block_types, mid_block_type, groups=32):
        super().__init__()
        if mid_block_type != "UNetMidBlock2D":
            raise NotImplementedError(mid_block_type)
        self.model_channels = model_channels
        temb = model_channels * 4
        self.time_embedding = TimestepEmbedding(model_channels, temb)
        self.conv_in = Conv2d(in_channels, model_channels, 3, padding=1)
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
        for m in self.modules():                      # zero-conv init (controller.py:174-185)
            if isinstance(m, ResnetBlock2D):
                nn.init.zeros_(m.conv2.weight), nn.init.zeros_(m.conv2.bias)
            if hasattr(m, "to_out") and hasattr(m, "group_norm"):
                nn.init.zeros_(m.to_out[0].weight), nn.init.zeros_(m.to_out[0].bias)

    def set_timesteps(self, timesteps):
        """(Re)build the per-resnet time-embedding bias tables.  Bumps `table_epoch`: an owning DiffUIE sees that its own
        schedule tables (and every graph captured against their addresses) are stale and rebuilds them."""
        emb = self.time_embedding.silu_emb(sinusoid_table(timesteps, self.model_channels))
        for r in _resnets(self):
            r.set_time_table(emb)
        self.table_epoch = getattr(self, "table_epoch", 0) + 1

    def stem(self, z_bf16):
        """conv_in(z0) does not depend on t: computed once per image, reused by every step (controller.py:198)."""
        return ops.conv(z_bf16, self.conv_in.packed(), gn=True)

    def run(self, stem, step):
        feats, h = [], stem
        for blk in self.down_blocks:
            states = []
            for i, res in enumerate(blk.resnets):
                h = res.run(h, step=step)
                if blk.attn_kind == "self":
                    h = blk.attentions[i].run(h)
                states.append(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0].run(h)
                states.append(h)
            feats.append(states[-2])                                   # output[-2] (controller.py:205)
        feats[-1] = self.middle_block.run(h, step=step)                # replace the last one (controller.py:211)
        return {f.shape[2]: self.fea_tran[i].run(f, step=step) for i, f in enumerate(feats)}   # keyed by WIDTH

    def run_schedule(self, stem, nsteps):
        """All `nsteps` evaluations of the schedule in ONE batched pass.  control_i = Controller(z0, t_i) never sees zt
        (unifie.py:148), so the S evaluations are independent: stacking them step-major into a batch of S*B images turns
        the small 16x16 / 8x8 levels into full MFMA tiles and 20 launches per layer into one.  Per-step time embeddings
        enter as per-image bias rows.  Returns the list of per-step control dicts (views of the batched outputs)."""
        b = stem.shape[0]
        x = stem.repeat(nsteps, 1, 1, 1)
        g = ops.gn_of(stem)
        if g is not None:                                  # the partial statistics repeat with the images
            x._gn = (g[0].repeat(nsteps, 1, 1, 1), g[1])
        out = self.run(x, "all")
        return [{k: v[i * b:(i + 1) * b] for k, v in out.items()} for i in range(nsteps)]

    def forward(self, x, timesteps, encoder_hidden_states=None):
        """Reference signature: x (B,4,h,w) fp32 NCHW, timesteps (1,) or (B,) -> {width: (B,256,h',w') fp32}."""
        _use_dtype(self)
        ts = _per_sample_timesteps(timesteps, x.shape[0])
        self.set_timesteps(ts)
        # one table row per distinct request: a (1,) tensor is row 0 for every image; a (B,) tensor gives image i row i
        # (controller.py:193-194 broadcasts the same way) - the step-major "all" form with one image per row
        out = self.run(self.stem(ops.nchw_to_nhwc(x.to(DEV))), 0 if len(ts) == 1 else "all")
        out = {k: ops.nhwc_to_nchw(v) for k, v in out.items()}
        _check_fp16(out)
        return out


class ControlledUNet(nn.Module):
    def __init__(self, unet: UNet2DConditionModel, control_type: str, null_embeds: Optional[torch.Tensor] = None,
                 cond_channels: int = 256):
        super().__init__()
        self.unet = unet
        cross_dim = unet.down_blocks[0].attentions[0].transformer_blocks[0].attn2.to_k.in_features
        if null_embeds is None:
            p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "sd_null_emb.pt")
            null_embeds = torch.load(p, map_location="cpu") if (cross_dim == 1024 and os.path.exists(p)) else \
                torch.zeros(1, 77, cross_dim)
        self.register_buffer("null_embeds", null_embeds.float())
        self.control_type = control_type
        if control_type == "spade":                      # base_model.py:32-37: a SPADE on every ResnetBlock2D of the UNet
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

    def set_timesteps(self, timesteps):
        u = self.unet
        emb = u.time_embedding.silu_emb(sinusoid_table(timesteps, u.time_proj_dim))
        for r in _resnets(u):
            r.set_time_table(emb)
        self.table_epoch = getattr(self, "table_epoch", 0) + 1          # see Controller.set_timesteps

    def _ctx(self):
        key = ("ctx", ops.act_dtype())
        if key not in self.__dict__:
            self.__dict__[key] = self.null_embeds.to(DEV, ops.act_dtype()).contiguous()
        return self.__dict__[key]

    def run(self, zt_bf16, control, step):
        """zt_bf16 [B,h,w,8] (latent channels zero-padded), control {width: NHWC bf16} -> eps fp32 [B,h,w,8]."""
        u, ctx = self.unet, self._ctx()
        h = ops.conv(zt_bf16, u.conv_in.packed(), gn=True)
        skips = [h]
        sp = control if self.control_type == "spade" else None    # spade_resnet vs _resnet (base_model.py:47-92)
        for blk in u.down_blocks:
            for i, res in enumerate(blk.resnets):
                h = res.run(h, step=step, control=sp)
                if blk.attn_kind == "cross":
                    h = blk.attentions[i].run(h, ctx)
                skips.append(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0].run(h)
                skips.append(h)
        # SC-Tuner (base_model.py:233-238): every skip is edited under the control feature of its resolution.  (A parallel graph
        # branch for the adapters was measured in round 2 - no gain once the GEMM ring race was fixed - and removed in round 3.)
        raw = skips
        if sp is not None:                                  # SPADE control: no skip editors (base_model.py:233 hasattr check)
            h = u.mid_block.run(h, step=step, ctx=ctx, control=sp)
            edited = raw
        else:
            h = u.mid_block.run(h, step=step, ctx=ctx)
            edited = [ed.run(s, control[s.shape[2]]) for ed, s in zip(self.csc_editors, raw)]
        idx = len(raw)
        for blk in u.up_blocks:
            for i, res in enumerate(blk.resnets):
                idx -= 1
                h = res.run(h, x2=edited[idx], step=step, control=sp)                           # virtual torch.cat
                if blk.attn_kind == "cross":
                    h = blk.attentions[i].run(h, ctx)
            if blk.upsamplers is not None:
                h = blk.upsamplers[0].run(h)
        h = u.conv_norm_out.run(h, silu=True)
        return ops.conv(h, u.conv_out.packed(), out_f32=True)

    def forward(self, sample, control, timesteps):
        """Reference signature (base_model.py:211-245): NCHW fp32 in, eps NCHW fp32 out."""
        _use_dtype(self)
        ts = _per_sample_timesteps(timesteps, sample.shape[0])
        self.set_timesteps(ts)
        ctl = {k: ops.nchw_to_nhwc(v.to(DEV)) for k, v in control.items()}
        eps = self.run(ops.nchw_to_nhwc(sample.to(DEV)), ctl, 0 if len(ts) == 1 else "all")     # (B,) timesteps: image i = bias row i
        eps = ops.nhwc_to_nchw(eps, c=self.unet.conv_out.out_channels)
        _check_fp16(eps)
        return eps


class SkipConnectedAutoEncoder(nn.Module):
    def __init__(self, vae: AutoencoderKL, fr_type: Optional[str] = None, tedit: Optional[dict] = None, fr_depths=(1, 1, 9)):
        super().__init__()
        self.tedit_dict = tedit
        self.vae = vae
        enc_ch = [b.resnets[-1].conv2.out_channels for b in vae.encoder.down_blocks]
        if fr_type == "CFRM":
            vae.encoder.fr_blocks = cfrm_blocks(enc_ch[:3], fr_depths)
        elif fr_type is not None:
            raise ValueError("Invalid fr_type")
        if tedit:
            self.task_list = list(tedit["task"])
            self.tedit_type = tedit["type"]
            if self.tedit_type != "TFA":
                raise KeyError("%s is not defined in the taskeditor!, please select ['TFA']" % self.tedit_type)
            pl, top = tedit["prompt_len"], enc_ch[-1]
            vae.decoder.task_prompts = nn.ParameterDict({t: nn.Parameter(torch.zeros(pl, enc_ch[2])) for t in self.task_list})
            vae.decoder.task_editors = nn.ModuleList([
                TaskFeatureAdapter(top, enc_ch[2], prompt_len=pl),
                TaskFeatureAdapter(top, enc_ch[1], prompt_len=pl),
                TaskFeatureAdapter(top, enc_ch[0], prompt_len=pl, last_layer=True)])
        else:
            self.task_list, self.tedit_type = [], None

    # ---- NHWC fast path -------------------------------------------------------------------------------------
    def encode_run(self, images_dev: torch.Tensor, noise_nchw: torch.Tensor, enable_fr: bool, plan=None):
        """images fp32 NCHW in [0,1] on device -> (z fp32 [B,h,w,8], z bf16, [3 NHWC bf16 skip features]).
        plan = (resized_h, resized_w, pad_h, pad_w): DiffUIE.forward's bicubic resize + reflect pad run in the layout kernel."""
        enc, lat = self.vae.encoder, self.vae.latent_channels
        x0 = ops.image_resize_pad(images_dev, *plan) if plan else ops.nchw_to_nhwc(images_dev, image=True)   # x*2-1 fused
        h = ops.conv(x0, enc.conv_in.packed(), gn=True)
        res = []
        for i, blk in enumerate(enc.down_blocks[:-1]):
            h = blk.run(h)
            if enable_fr:
                h = enc.fr_blocks[i].run(h)
            res.append(h)
        h = enc.mid_block.run(enc.down_blocks[-1].run(h))
        h = ops.conv(enc.conv_norm_out.run(h, silu=True), enc.conv_out.packed())
        moments = ops.conv(h, self.vae.quant_conv.packed(), out_f32=True)
        z, zb = ops.vae_sample(moments, noise_nchw, lat, self.vae.config.scaling_factor)
        return z, zb, res

    def decode_run(self, z_f32: torch.Tensor, res_samples, task: str, out_plan=None):
        """out_plan = (crop_hw, out_hw, quantize): un-pad + bicubic resize back (+ 8-bit quantisation) in the layout kernel."""
        dec, lat = self.vae.decoder, self.vae.latent_channels
        if self.tedit_dict and task not in dec.task_prompts:
            raise KeyError(task)
        zb = ops.f32_to_bf16(z_f32, lat, mul=1.0 / self.vae.config.scaling_factor)
        h = ops.conv(ops.conv(zb, self.vae.post_quant_conv.packed()), dec.conv_in.packed(), gn=True)
        h = dec.mid_block.run(h)
        if not self.tedit_dict:                                # stock VAE decoder: the reference only patches the decoder
            for blk in dec.up_blocks:                          # forward when a task editor is configured (autoencoder.py:107-110)
                h = blk.run(h)
        else:
            b = z_f32.shape[0]
            key = ("cache", "prompt", task)
            if key not in self.__dict__:                       # device copy made once (not inside graph capture)
                self.__dict__[key] = dec.task_prompts[task].detach().float().to(DEV)
            cond = self.__dict__[key].unsqueeze(0).expand(b, -1, -1).contiguous()
            for i, blk in enumerate(dec.up_blocks[:-1]):
                h, cond = dec.task_editors[i].run(h, res_samples[-i - 1], cond)
                h = blk.run(h)
            h = dec.up_blocks[-1].run(h)
        h = ops.conv(dec.conv_norm_out.run(h, silu=True), dec.conv_out.packed(), out_f32=True)
        if out_plan:
            return ops.image_unpad_resize(h, dec.conv_out.out_channels, out_plan[0], out_plan[1], mul=0.5, add=0.5, quantize=out_plan[2])
        return ops.nhwc_to_nchw(h, c=dec.conv_out.out_channels, mul=0.5, add=0.5)             # (x+1)/2

    # ---- reference signatures ---------------------------------------------------------------------------------
    def encode(self, images, enable_fr: bool = False, noise=None):
        _use_dtype(self)
        images = images.to(DEV).float()
        b, _, hh, ww = images.shape
        if noise is None:
            noise = torch.randn(b, self.vae.latent_channels, hh // 8, ww // 8, device=DEV)
        z, _, res = self.encode_run(images, noise.to(DEV).float().contiguous(), enable_fr)
        return ops.nhwc_to_nchw(z, c=self.vae.latent_channels), [ops.nhwc_to_nchw(r) for r in res]

    def decode(self, latents, res_samples, task: str):
        _use_dtype(self)
        z = ops.nchw_to_nhwc(latents.to(DEV)).float()
        return self.decode_run(z.contiguous(), [ops.nchw_to_nhwc(r.to(DEV)) for r in res_samples], task)

    def forward(self, images, task: str):
        latents, res = self.encode(images, enable_fr=True)
        return self.decode(latents, res, "ir")


def resize_pad_plan(h: int, w: int):
    """Integer shape arithmetic of unifie.py:121-134 -> (resized_h, resized_w, pad_h, pad_w)."""
    if h < 512 or w < 512:
        s = 512 / min(h, w)
        h, w = round(h * s), round(w * s)
    return h, w, (64 - h % 64) % 64, (64 - w % 64) % 64


class DiffUIE(nn.Module):
    """forward(images, task) -> restored images, same contract as the reference (fp32 NCHW in [0,1])."""

    def __init__(self, frenc: Optional[dict] = None, cnet: Optional[dict] = None, tedit: Optional[dict] = None, *,
                 unet_cfg=None, vae_cfg=None, controller_cfg=None, null_embeds=None, fr_depths=(1, 1, 9), use_graph=True,
                 dtype="bf16"):
        """dtype: "bf16" (the reference's bf16-mixed precision) or "fp16" - the 16-bit type activations and weights are
        stored in and fed to the matrix cores; accumulation, statistics, softmax and the DDIM state are fp32 in both."""
        super().__init__()
        self.dtype = ops.set_dtype(dtype)
        self.fr_type = frenc["type"] if frenc else None
        self.control_type = cnet["type"] if cnet else None
        self.tedit = tedit if tedit else None
        self.ae = SkipConnectedAutoEncoder(AutoencoderKL(**(vae_cfg or {})), self.fr_type, self.tedit, fr_depths)
        self.use_graph = use_graph
        self.trace_zt = None                   # parity instrumentation: a list collects zt after every DDIM step (eager runs only)
        self.check_fp16_overflow = True        # fp16 only: one isfinite reduction over the restored images per forward (+ a sync)
        self.batch_controller = os.environ.get("UR_BATCH_CONTROLLER", "1") == "1"
        self._graphs = {}
        if self.control_type:
            ccfg = controller_cfg or stablesr_config
            self.controller = Controller(**ccfg)
            self.base_model = ControlledUNet(UNet2DConditionModel(**(unet_cfg or {})), self.control_type, null_embeds,
                                             ccfg["out_channels"])
            self.register_buffer("train_timesteps", torch.tensor([249, 499, 749, 999, 999, 999], dtype=torch.int64))
            self.num_inference_steps = int(cnet["num_inference_steps"])
            self.timesteps = schedule.ddim_timesteps(self.num_inference_steps)       # host int64, bit-exact
            self._tables_ready = False
            self._table_epochs = None
        self._own_dtype()
        self._arm_load_hooks()

    def _own_dtype(self):
        """The operator-level entry points of the sub-modules (ae.encode / decode, Controller.forward, ControlledUNet.forward) run
        in THIS model's 16-bit type, whatever another model set in between."""
        for m in (self.ae, getattr(self, "controller", None), getattr(self, "base_model", None)):
            if m is not None:
                m.__dict__["_ur_dtype"] = self.dtype
                m.__dict__["_ur_owner"] = weakref.ref(self)

    def _arm_load_hooks(self):
        """Every module of the tree reports a finished `load_state_dict` (the reference's engine loads SUB-module state dicts:
        engine_unifie.py:58,75,81,114,125): the packed device copies, schedule tables and captured graphs derived from the fp32
        masters are then rebuilt before the next use - no `refresh()` call is needed at the integration site."""
        ref = weakref.ref(self)

        def loaded(_module, _incompatible):
            o = ref()
            if o is not None:
                o.__dict__["_stale"] = True
        for m in self.modules():
            m.register_load_state_dict_post_hook(loaded)

    def set_num_inference_steps(self, n: int):
        """Change the DDIM schedule length (`cnet.num_inference_steps`, unifie.py:70-75): tables and graphs are rebuilt lazily."""
        self.num_inference_steps = int(n)
        self.timesteps = schedule.ddim_timesteps(self.num_inference_steps)
        self._tables_ready = False
        return self

    def set_dtype(self, dtype):
        """Switch the 16-bit compute type; packed weights are kept per type, captured graphs are dropped."""
        self.dtype = ops.set_dtype(dtype)
        self._own_dtype()
        self._graphs.clear()
        return self

    # ---- weights ------------------------------------------------------------------------------------------------
    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.refresh()
        return r

    def refresh(self):
        """Drop device copies derived from the fp32 masters (after loading / editing weights)."""
        invalidate_packed(self)
        self._tables_ready = False
        self._graphs.clear()
        self.__dict__["_stale"] = False

    def _prepare(self):
        if self.__dict__.get("_stale"):
            self.refresh()
        ops.set_dtype(self.dtype)
        if not self.control_type:
            return
        # The schedule's bias tables belong to the resnets; an ad-hoc Controller.forward / ControlledUNet.forward /
        # predict_z0 call rebinds them (table_epoch moves).  Graphs captured against the old tables hold dangling addresses:
        # rebuild the tables and drop the graphs whenever the epochs are not the ones this object last set.
        epochs = (getattr(self.controller, "table_epoch", 0), getattr(self.base_model, "table_epoch", 0))
        if not self._tables_ready or epochs != self._table_epochs:
            self._graphs.clear()
            self.controller.set_timesteps(self.timesteps)
            self.base_model.set_timesteps(self.timesteps)
            self._table_epochs = (self.controller.table_epoch, self.base_model.table_epoch)
            self._tables_ready = True

    # ---- reference helper signatures ------------------------------------------------------------------------------
    def diffuse(self, latents, timesteps=None, noise=None):
        if self.__dict__.get("_stale"):
            self.refresh()
        ops.set_dtype(self.dtype)
        latents = latents.to(DEV).float()
        if timesteps is None:
            timesteps = self.train_timesteps[torch.randint(0, len(self.train_timesteps), (latents.size(0),))]
        ts = [int(t) for t in torch.as_tensor(timesteps).reshape(-1).tolist()]
        noise = torch.randn_like(latents) if noise is None else noise.to(DEV).float()
        ac = schedule.alphas_cumprod()
        lat = latents.shape[1]
        # fp32 throughout (the latent state never passes through a 16-bit tensor): NHWC fp32 padded to 8 channels for the kernel
        z = F.pad(latents.permute(0, 2, 3, 1), (0, 8 - lat)).contiguous()
        outs = []
        for i, t in enumerate(ts):       # per-sample t (training helper); inference uses one t for the batch
            a_t = np.float32(float(ac[t]))
            sa, sb = float(np.sqrt(a_t)), float(np.sqrt(np.float32(1) - a_t))          # fp32 sqrt, as DDPMScheduler.add_noise
            zt, _ = ops.add_noise(z[i:i + 1].contiguous(), noise[i:i + 1].contiguous(), lat, sa, sb)
            outs.append(ops.nhwc_to_nchw(zt, c=lat))
        return torch.cat(outs, 0), noise, torch.as_tensor(ts)

    # ---- the hot path ------------------------------------------------------------------------------------------------
    def _forward_device(self, images, task, n_vae, n_t, plan, quantize=False):
        """images fp32 NCHW on device (original size); plan = resize_pad_plan(H, W).
        Returns (preds NCHW fp32 at the original size, z0, zt) (NHWC fp32 latents)."""
        lat = self.ae.vae.latent_channels
        h, w, pad_h, pad_w = plan
        z0, z0b, mids = self.ae.encode_run(images, n_vae, enable_fr=self.fr_type is not None, plan=plan)
        zt = z0
        if self.control_type:
            ac = schedule.alphas_cumprod_f64()
            zt, ztb = ops.add_noise(z0, n_t, lat, float(np.float32(ac[999] ** 0.5)), float(np.float32((1 - ac[999]) ** 0.5)))
            stem = self.controller.stem(z0b)
            controls = self.controller.run_schedule(stem, len(self.timesteps)) if self.batch_controller else None
            for i, t in enumerate(self.timesteps):
                control = controls[i] if controls is not None else self.controller.run(stem, i)
                eps = self.base_model.run(ztb, control, i)
                c_x, c_e = schedule.ddim_coefficients(int(t), self.num_inference_steps)
                ops.ddim_step_(zt, ztb, eps, lat, c_x, c_e)
                if self.trace_zt is not None and not torch.cuda.is_current_stream_capturing():
                    self.trace_zt.append(ops.nhwc_to_nchw(zt, c=lat).cpu())      # parity instrumentation (eager runs only)
        preds = self.ae.decode_run(zt, mids, task, out_plan=((h, w), tuple(images.shape[-2:]), quantize))
        return preds, z0, zt

    @torch.no_grad()
    def forward(self, images, task: str, noise=None, return_latents=False, quantize=False):
        """noise = (eps_vae, eps_t999): the two RNG draws of the reference (autoencoder.py:152, unifie.py:87), NCHW fp32.
        quantize=True additionally applies the evaluator's mul(255).round().clamp(0,255).div(255) (eval_image_restoration.py:71).
        Resize / reflect pad / un-pad / resize back (unifie.py:124-134,164-168) run as HIP kernels inside the graph."""
        if task not in self.ae.task_list and self.tedit:
            raise KeyError(task)
        self._prepare()
        images = images.to(DEV).float().contiguous()
        org_h, org_w = images.shape[-2:]
        plan = resize_pad_plan(org_h, org_w)
        h, w, pad_h, pad_w = plan
        b, lat = images.shape[0], self.ae.vae.latent_channels
        lh, lw = (h + pad_h) // 8, (w + pad_w) // 8
        if noise is None:
            noise = (torch.randn(b, lat, lh, lw, device=DEV), torch.randn(b, lat, lh, lw, device=DEV))
        n_vae, n_t = (n.to(DEV).float().contiguous() for n in noise)
        if tuple(n_vae.shape) != (b, lat, lh, lw) or tuple(n_t.shape) != (b, lat, lh, lw):
            raise ValueError(f"noise must be two tensors of shape {(b, lat, lh, lw)}")
        if self.use_graph:
            preds, z0, zt = self._graph_forward(images, task, n_vae, n_t, plan, quantize)
        else:
            preds, z0, zt = self._forward_device(images, task, n_vae, n_t, plan, quantize)
        if self.dtype == torch.float16 and self.check_fp16_overflow and not bool(torch.isfinite(preds).all()):
            # fp16 conversions overflow to inf (csrc/common.h Act<true>::pack2); an inf becomes NaN in the next GroupNorm /
            # softmax and reaches the image.  Heavy-tailed activations (real SD-2.x weights can produce them) need bf16.
            raise FloatingPointError("fp16 activation overflow (|x| > 65504) somewhere in the forward: the restored image is not "
                                     "finite.  Run this model with dtype='bf16' (DiffUIE.set_dtype('bf16') / trainer.precision: bf16-mixed)")
        if return_latents:
            return preds, ops.nhwc_to_nchw(z0, c=lat), ops.nhwc_to_nchw(zt, c=lat)
        return preds

    # ---- hipGraph: the whole fixed-length forward (encode, N denoise steps, decode) is one captured graph --------------
    def _graph_forward(self, images, task, n_vae, n_t, plan, quantize=False):
        key = (tuple(images.shape), task, bool(quantize), self.dtype)
        g = self._graphs.get(key)
        if g is None:
            static = dict(images=images.clone(), n_vae=n_vae.clone(), n_t=n_t.clone())
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                       # warm-up: packs weights, sizes workspaces, sets func attributes
                self._forward_device(static["images"], task, static["n_vae"], static["n_t"], plan, quantize)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):    # an RCCL watchdog thread may be alive
                outs = self._forward_device(static["images"], task, static["n_vae"], static["n_t"], plan, quantize)
            max_graphs = int(os.environ.get("UR_MAX_GRAPHS", "8"))      # each captured shape keeps its own activation pool
            while len(self._graphs) >= max(max_graphs, 1):
                self._graphs.pop(next(iter(self._graphs)))                # oldest first
            g = self._graphs[key] = (graph, static, outs)
        graph, static, outs = g
        static["images"].copy_(images)
        static["n_vae"].copy_(n_vae)
        static["n_t"].copy_(n_t)
        if os.environ.get("UR_DEBUG_SYNC"):
            torch.cuda.synchronize()
        graph.replay()
        if os.environ.get("UR_DEBUG_SYNC"):
            torch.cuda.synchronize()
        # the graph's output tensors are overwritten by the next replay of this (shape, task) graph: hand the caller copies
        # (runner.forward keeps [enh_hq, enh_lq] of two same-shape calls; a copy is tiny next to a forward)
        return tuple(o.clone() for o in outs)

    def predict_z0(self, latents, conditions, timesteps):
        """unifie.py:91-105 (training-side helper): per-sample timesteps via per-image bias rows."""
        if self.__dict__.get("_stale"):
            self.refresh()
        ops.set_dtype(self.dtype)
        lat = latents.shape[1]
        ts = [int(t) for t in torch.as_tensor(timesteps).reshape(-1).tolist()]
        if len(ts) == 1:
            ts = ts * latents.shape[0]
        self.controller.set_timesteps(ts)             # bumps the table epochs: the next forward() rebuilds the schedule tables
        self.base_model.set_timesteps(ts)             # and drops its graphs (see _prepare)
        zb = ops.nchw_to_nhwc(latents.to(DEV))
        cb = ops.nchw_to_nhwc(conditions.to(DEV))
        outs = []
        for i, t in enumerate(ts):
            control = self.controller.run(self.controller.stem(cb[i:i + 1].contiguous()), i)
            eps = ops.nhwc_to_nchw(self.base_model.run(zb[i:i + 1].contiguous(), control, i), c=lat)
            a = float(schedule.alphas_cumprod()[t])
            outs.append((latents[i:i + 1].to(DEV).float() - (1 - a) ** 0.5 * eps) / a ** 0.5)
        z0 = torch.cat(outs, 0)
        _check_fp16(z0)
        return z0
