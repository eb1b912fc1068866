"""Weight ingest on CPU: synthetic files in the reference's on-disk layouts (HF safetensors folders,
Lightning `state_dict` checkpoints with `model.`-prefixed keys) must land in the right parameters, strictness as in
engine_unifie.py:51-126."""
import os

import pytest
import torch

from tiny_cfg import TINY, model_kwargs, randomise_


def _tiny():
    from unirestore_amd.modules import DiffUIE
    return DiffUIE(**model_kwargs(1), **TINY)


def test_lightning_prefix_slices_and_hf_folders(tmp_path):
    from safetensors.torch import save_file
    from unirestore_amd import checkpoint as ck
    torch.manual_seed(0)
    src = _tiny(); randomise_(src)
    full = {"model." + k: v.clone() for k, v in src.state_dict().items()}
    full["some.other.module.weight"] = torch.zeros(3)                         # Lightning checkpoints hold unrelated keys too
    for name in ("frenc", "cnet", "tedit"):
        torch.save({"state_dict": full, "epoch": 1}, tmp_path / f"{name}.ckpt")
    os.makedirs(tmp_path / "hf" / "unet"); os.makedirs(tmp_path / "hf" / "vae")
    save_file({k: v.contiguous() for k, v in src.base_model.unet.state_dict().items()},
              str(tmp_path / "hf" / "unet" / "diffusion_pytorch_model.safetensors"))
    vae_sd = {k: v.contiguous() for k, v in src.ae.vae.state_dict().items() if not k.startswith(ck.ADAPTER_KEYS)}
    save_file(vae_sd, str(tmp_path / "hf" / "vae" / "diffusion_pytorch_model.safetensors"))

    kw = model_kwargs(1)
    kw["frenc"]["ckpt_path"] = str(tmp_path / "frenc.ckpt")
    kw["cnet"]["ckpt_path"] = str(tmp_path / "cnet.ckpt")
    kw["tedit"]["ckpt_path"] = str(tmp_path / "tedit.ckpt")
    dst = ck.build_from_config(kw, hf_root=str(tmp_path / "hf"), **TINY)
    torch.save(src.base_model.null_embeds.clone(), tmp_path / "sd_null_emb.pt")     # shipped as its own file (base_model.py:23-27)
    ck.load_null_embeds(dst, str(tmp_path / "sd_null_emb.pt"))
    a, b = src.state_dict(), dst.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert not any(p.requires_grad for p in dst.parameters())

    # reference slices: k[31:], k[17:], k[29:]
    assert len(ck.PREFIX_FR) == 31 and len(ck.PREFIX_CONTROLLER) == 17 and len(ck.PREFIX_CSC) == 29
    # strictness: a controller checkpoint with a missing key must fail, prompts with a missing task must not
    broken = {k: v for k, v in full.items() if k != "model.controller.conv_in.weight"}
    torch.save({"state_dict": broken}, tmp_path / "broken.ckpt")
    with pytest.raises(RuntimeError):
        ck.load_adapter_checkpoints(dst, cnet={"ckpt_path": str(tmp_path / "broken.ckpt")})
    fewer = {k: v for k, v in full.items() if not k.startswith(ck.PREFIX_PROMPTS + "seg")}
    torch.save({"state_dict": fewer}, tmp_path / "fewer.ckpt")
    ck.load_adapter_checkpoints(dst, tedit={"ckpt_path": str(tmp_path / "fewer.ckpt")})
    with pytest.raises(KeyError):
        torch.save({"weights": {}}, tmp_path / "notlightning.ckpt")
        ck.load_adapter_checkpoints(dst, frenc={"ckpt_path": str(tmp_path / "notlightning.ckpt")})
    # VAE file with a foreign key is rejected; a truncated UNet file is rejected (strict)
    save_file({**vae_sd, "bogus.weight": torch.zeros(1)}, str(tmp_path / "hf" / "vae" / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(RuntimeError):
        ck.load_hf_weights(dst, str(tmp_path / "hf"), components=("vae",))
    with pytest.raises(FileNotFoundError):
        ck.load_hf_weights(dst, str(tmp_path / "nowhere"))
    with pytest.raises(ValueError):
        torch.save(torch.zeros(1, 5, 7), tmp_path / "null.pt")
        ck.load_null_embeds(dst, str(tmp_path / "null.pt"))
    torch.save(torch.full(tuple(dst.base_model.null_embeds.shape), 0.5), tmp_path / "null_ok.pt")
    ck.load_null_embeds(dst, str(tmp_path / "null_ok.pt"))
    assert float(dst.base_model.null_embeds.mean()) == 0.5


def test_spade_checkpoint_slices(tmp_path):
    """control_type 'spade': controller + the SPADE modules inside base_model.unet come out of the cnet checkpoint."""
    from unirestore_amd import checkpoint as ck
    from unirestore_amd.modules import DiffUIE
    kw = model_kwargs(1)
    kw["cnet"]["type"] = "spade"
    torch.manual_seed(1)
    src = DiffUIE(**kw, **TINY); randomise_(src, 4)
    torch.save({"state_dict": {"model." + k: v.clone() for k, v in src.state_dict().items()}}, tmp_path / "cnet.ckpt")
    dst = DiffUIE(**kw, **TINY)
    kw["cnet"]["ckpt_path"] = str(tmp_path / "cnet.ckpt")
    ck.load_adapter_checkpoints(dst, cnet=kw["cnet"])
    a, b = src.state_dict(), dst.state_dict()
    n = 0
    for k in a:
        if ".spade." in k or k.startswith("controller."):
            assert torch.equal(a[k], b[k]), k
            n += 1
    assert n > 100 and not hasattr(dst.base_model, "csc_editors")


def test_hf_unet_into_spade_model_and_deprecated_vae_names(tmp_path):
    """ An HF UNet file has no SPADE keys: a control_type='spade' model must still take it (only '.spade.' keys may
    be missing); SD-2.x VAE exports with the pre-0.18 attention names (query / key / value / proj_attn) load after the rename."""
    from safetensors.torch import save_file
    from unirestore_amd import checkpoint as ck
    from unirestore_amd.modules import DiffUIE
    kw = model_kwargs(1)
    torch.manual_seed(2)
    plain = DiffUIE(**kw, **TINY); randomise_(plain, 5)
    os.makedirs(tmp_path / "hf" / "unet"); os.makedirs(tmp_path / "hf" / "vae")
    save_file({k: v.contiguous() for k, v in plain.base_model.unet.state_dict().items()}, str(tmp_path / "hf" / "unet" / "diffusion_pytorch_model.safetensors"))
    old = {}
    for k, v in plain.ae.vae.state_dict().items():
        if k.startswith(("encoder.fr_blocks.", "decoder.task_")):
            continue
        for new, dep in ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn.")):
            if ".attentions." in k and new in k:
                k = k.replace(new, dep)
        old[k] = v.contiguous()
    assert any(".proj_attn." in k for k in old)
    save_file(old, str(tmp_path / "hf" / "vae" / "diffusion_pytorch_model.safetensors"))
    kw2 = model_kwargs(1); kw2["cnet"]["type"] = "spade"
    dst = DiffUIE(**kw2, **TINY)
    rep = ck.load_hf_weights(dst, str(tmp_path / "hf"))
    assert rep["unet"] > 100 and rep["vae"] > 100
    a, b = plain.base_model.unet.state_dict(), dst.base_model.unet.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a) and any(".spade." in k for k in b)
    va, vb = plain.ae.vae.state_dict(), dst.ae.vae.state_dict()
    assert all(torch.equal(va[k], vb[k]) for k in va if ".attentions." in k)
    # a UNet file that lacks a NON-spade key is still rejected
    bad = {k: v.contiguous() for k, v in plain.base_model.unet.state_dict().items() if k != "conv_in.bias"}
    save_file(bad, str(tmp_path / "hf" / "unet" / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(RuntimeError):
        ck.load_hf_weights(dst, str(tmp_path / "hf"), components=("unet",))


def test_per_sample_timestep_contract():
    """Operator-level timesteps (controller.py:193-194, base_model.py:211-216): one value for the batch or one per sample."""
    import pytest
    import torch
    from unirestore_amd.modules.model import _per_sample_timesteps
    assert _per_sample_timesteps(torch.tensor([999]), 4) == [999]
    assert _per_sample_timesteps(torch.tensor(749), 3) == [749]
    assert _per_sample_timesteps(torch.tensor([5, 5, 5]), 3) == [5]                  # all equal collapses to the shared-row form
    assert _per_sample_timesteps(torch.tensor([999, 249, 749]), 3) == [999, 249, 749]
    with pytest.raises(ValueError):
        _per_sample_timesteps(torch.tensor([999, 249]), 3)
