"""Weight ingest for the hot path: the on-disk formats the reference reads before `DiffUIE.forward`.

* Hugging Face layout of `stabilityai/sd-turbo` (reference unifie.py:40,60 `from_pretrained(..., subfolder="vae"|"unet")`):
  `<root>/unet/diffusion_pytorch_model[.fp16].safetensors` and `<root>/vae/...` - parameter names are the diffusers
  names, which the module trees here reproduce one-to-one, so ingest is a strict `load_state_dict`.
* Lightning checkpoints of the three adapter stages (engine_unifie.py:51-58, 68-81, 104-126): `torch.load(path)["state_dict"]`
  sliced by attribute-path prefix.
* `sd_null_emb.pt` (base_model.py:23-27): the (1, 77, 1024) empty-prompt embedding.

Everything lands in the fp32 master parameters; `model.refresh()` then drops the packed bf16 device copies so the next
forward re-packs them (and re-captures its graph).
"""
import os
from typing import Dict, Iterable, Optional

import torch

PREFIX_FR = "model.ae.vae.encoder.fr_blocks."
PREFIX_CONTROLLER = "model.controller."
PREFIX_CSC = "model.base_model.csc_editors."
PREFIX_PROMPTS = "model.ae.vae.decoder.task_prompts."
PREFIX_EDITORS = "model.ae.vae.decoder.task_editors."


def slice_prefix(state: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    """{k[len(prefix):]: v for k in state if k.startswith(prefix)} - the reference's `k[31:]` / `k[17:]` / `k[29:]` slices."""
    return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}


def _lightning_state(path: str) -> Dict[str, torch.Tensor]:
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "state_dict" not in ckpt:
        raise KeyError(f"{path}: not a Lightning checkpoint (no 'state_dict')")
    return ckpt["state_dict"]


def load_adapter_checkpoints(model, frenc: Optional[dict] = None, cnet: Optional[dict] = None, tedit: Optional[dict] = None):
    """Step 2 of `LitUniFIE.configure_model` for the `model_kwargs.{frenc,cnet,tedit}` dicts of configs/val.yaml.
    Strictness follows the reference: everything strict except the task prompts (`strict=False`, so a checkpoint
    trained with fewer tasks still loads)."""
    loaded = []
    if frenc and frenc.get("ckpt_path"):
        sd = slice_prefix(_lightning_state(frenc["ckpt_path"]), PREFIX_FR)
        model.ae.vae.encoder.fr_blocks.load_state_dict(sd)
        loaded.append(("frenc", len(sd)))
    if cnet and cnet.get("ckpt_path"):
        state = _lightning_state(cnet["ckpt_path"])
        sd = slice_prefix(state, PREFIX_CONTROLLER)
        model.controller.load_state_dict(sd)
        if hasattr(model.base_model, "csc_editors"):
            sd2 = slice_prefix(state, PREFIX_CSC)
            model.base_model.csc_editors.load_state_dict(sd2)
        else:
            # control_type "spade": the trainable part of base_model are the SPADE modules grafted onto the UNet's resnets
            # (engine_unifie.py:91-95 trains every module whose name contains "spade"; the reference's loader has no branch
            # for them, so this follows the parameter names the training run would have saved)
            sd2 = {k: v for k, v in slice_prefix(state, "model.base_model.").items() if ".spade." in k}
            res = model.base_model.load_state_dict(sd2, strict=False)
            want = {k for k in model.base_model.state_dict() if ".spade." in k}
            if res.unexpected_keys or want - set(sd2):
                raise RuntimeError(f"SPADE weights do not match: unexpected {list(res.unexpected_keys)[:4]} "
                                   f"missing {sorted(want - set(sd2))[:4]}")
        loaded.append(("cnet", len(sd) + len(sd2)))
    if tedit and tedit.get("ckpt_path"):
        state = _lightning_state(tedit["ckpt_path"])
        sd = slice_prefix(state, PREFIX_PROMPTS)
        model.ae.vae.decoder.task_prompts.load_state_dict(sd, strict=False)
        sd2 = slice_prefix(state, PREFIX_EDITORS)
        model.ae.vae.decoder.task_editors.load_state_dict(sd2)
        loaded.append(("tedit", len(sd) + len(sd2)))
    model.refresh()
    return loaded


def _find_weights(folder: str) -> str:
    for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                 "diffusion_pytorch_model.bin", "diffusion_pytorch_model.fp16.bin"):
        p = os.path.join(folder, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no diffusion_pytorch_model.{{safetensors,bin}} under {folder}")


def read_tensors(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu", weights_only=True)


# SD-2.x VAE exports made before diffusers 0.18 name the mid-block attention projections query / key / value / proj_attn
# (diffusers renames them on load); newer ones use to_q / to_k / to_v / to_out.0
_DEPRECATED_ATTN = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}


def remap_deprecated_vae_attention(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if ".attentions." in k:
            for old, new in _DEPRECATED_ATTN.items():
                if old in k:
                    k = k.replace(old, new)
                    if v.dim() == 4 and v.shape[-2:] == (1, 1):      # very old exports hold them as 1x1 convs
                        v = v[:, :, 0, 0]
                    break
        out[k] = v
    return out


ADAPTER_KEYS = ("encoder.fr_blocks.", "decoder.task_prompts.", "decoder.task_editors.")


def load_hf_weights(model, root: str, components: Iterable[str] = ("unet", "vae")):
    """`<root>/{unet,vae}/diffusion_pytorch_model.*` -> fp32 masters.  The UNet must match exactly; for the VAE the only
    keys allowed to be absent from the file are the adapters UniRestore grafts onto it (CFRM / TFA / task prompts)."""
    report = {}
    if "unet" in components and getattr(model, "control_type", None):
        sd = read_tensors(_find_weights(os.path.join(root, "unet")))
        # control_type "spade" grafts SPADE modules onto the UNet's resnets BEFORE the weights arrive (the reference calls
        # from_pretrained first, base_model.py:32-37): those - and only those - may be absent from the HF file
        res = model.base_model.unet.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
        bad = [k for k in res.missing_keys if ".spade." not in k]
        if bad or res.unexpected_keys:
            raise RuntimeError(f"UNet weights do not match: missing {bad[:5]} unexpected {list(res.unexpected_keys)[:5]}")
        report["unet"] = len(sd)
    if "vae" in components:
        sd = remap_deprecated_vae_attention(read_tensors(_find_weights(os.path.join(root, "vae"))))
        res = model.ae.vae.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
        bad = [k for k in res.missing_keys if not k.startswith(ADAPTER_KEYS)]
        if bad or res.unexpected_keys:
            raise RuntimeError(f"VAE weights do not match: missing {bad[:5]} unexpected {list(res.unexpected_keys)[:5]}")
        report["vae"] = len(sd)
    model.refresh()
    return report


def load_null_embeds(model, path: str):
    """`sd_null_emb.pt`: tensor (1, 77, cross_attention_dim), the text-encoder output of the empty prompt."""
    t = torch.load(path, map_location="cpu", weights_only=True).float()
    if tuple(t.shape) != tuple(model.base_model.null_embeds.shape):
        raise ValueError(f"null embedding shape {tuple(t.shape)} != {tuple(model.base_model.null_embeds.shape)}")
    with torch.no_grad():
        model.base_model.null_embeds.copy_(t)
    model.refresh()


def build_from_config(model_kwargs: dict, hf_root: Optional[str] = None, **model_overrides):
    """configs/val.yaml `model.init_args.model_kwargs` -> a ready `DiffUIE` (steps 1-2 of configure_model)."""
    from .modules import DiffUIE
    frenc, cnet, tedit = model_kwargs.get("frenc"), model_kwargs.get("cnet"), model_kwargs.get("tedit")
    model = DiffUIE(frenc=frenc, cnet=cnet, tedit=tedit, **model_overrides)
    model.requires_grad_(False).eval()
    if hf_root:
        load_hf_weights(model, hf_root)
    load_adapter_checkpoints(model, frenc, cnet, tedit)
    return model
