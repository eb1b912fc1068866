"""Weight ingest end to end ON THE GPU: files in the reference's on-disk layouts - Hugging Face
`unet/` + `vae/` safetensors folders, Lightning `frenc / cnet / tedit` checkpoints with `model.`-prefixed keys, `sd_null_emb.pt` -
are read by `checkpoint.build_from_config` (engine_unifie.py:51-126, unifie.py:40,60), packed for the HIP kernels, and the
forward is compared with the ORACLE built from the SAME files.  (The CPU-side file handling is tests/test_checkpoint_cpu.py.)"""
import os

import pytest
import torch

from golden_util import rel_l2
from tiny_cfg import TINY, model_kwargs, randomise_

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [("bf16", 8.5e-3), ("fp16", 1.1e-3)])      # the tiny full-forward tolerances (test_modules_gpu.TOL)
def test_forward_from_files_matches_oracle_from_the_same_files(tmp_path, dtype, tol):
    from safetensors.torch import save_file
    from oracle.model import DiffUIE as ODiffUIE
    from unirestore_amd import checkpoint as ck
    torch.manual_seed(0)
    src = randomise_(ODiffUIE(**model_kwargs(2), **TINY).eval(), 21)         # the "published" weights
    full = {"model." + k: v.clone() for k, v in src.state_dict().items()}
    for name in ("frenc", "cnet", "tedit"):
        torch.save({"state_dict": full, "epoch": 3}, tmp_path / f"{name}.ckpt")
    os.makedirs(tmp_path / "hf" / "unet"); os.makedirs(tmp_path / "hf" / "vae")
    save_file({k: v.contiguous() for k, v in src.base_model.unet.state_dict().items()},
              str(tmp_path / "hf" / "unet" / "diffusion_pytorch_model.safetensors"))
    save_file({k: v.contiguous() for k, v in src.ae.vae.state_dict().items() if not k.startswith(ck.ADAPTER_KEYS)},
              str(tmp_path / "hf" / "vae" / "diffusion_pytorch_model.safetensors"))
    torch.save(src.base_model.null_embeds.clone(), tmp_path / "sd_null_emb.pt")
    kw = model_kwargs(2)
    for k in ("frenc", "cnet", "tedit"):
        kw[k]["ckpt_path"] = str(tmp_path / f"{k}.ckpt")

    # product: files -> fp32 masters -> packed 16-bit device copies -> HIP forward
    p = ck.build_from_config(kw, hf_root=str(tmp_path / "hf"), dtype=dtype, **TINY)
    ck.load_null_embeds(p, str(tmp_path / "sd_null_emb.pt"))
    # oracle: a FRESH oracle model fed from the same files through the same slicing rules (not from `src` directly)
    o = ODiffUIE(**model_kwargs(2), **TINY).eval()
    o.base_model.unet.load_state_dict(ck.read_tensors(str(tmp_path / "hf" / "unet" / "diffusion_pytorch_model.safetensors")))
    o.ae.vae.load_state_dict(ck.read_tensors(str(tmp_path / "hf" / "vae" / "diffusion_pytorch_model.safetensors")), strict=False)
    st = torch.load(tmp_path / "cnet.ckpt", weights_only=False)["state_dict"]
    o.ae.vae.encoder.fr_blocks.load_state_dict(ck.slice_prefix(st, ck.PREFIX_FR))
    o.controller.load_state_dict(ck.slice_prefix(st, ck.PREFIX_CONTROLLER))
    o.base_model.csc_editors.load_state_dict(ck.slice_prefix(st, ck.PREFIX_CSC))
    o.ae.vae.decoder.task_prompts.load_state_dict(ck.slice_prefix(st, ck.PREFIX_PROMPTS), strict=False)
    o.ae.vae.decoder.task_editors.load_state_dict(ck.slice_prefix(st, ck.PREFIX_EDITORS))
    with torch.no_grad():
        o.base_model.null_embeds.copy_(torch.load(tmp_path / "sd_null_emb.pt"))
    g = torch.Generator().manual_seed(4)
    img = torch.rand(2, 3, 64, 64, generator=g)
    noise = (torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 4, 64, 64, generator=g))
    ref, rz0, rzt = o(img, "cls", noise=noise, return_latents=True)
    out, z0, zt = p(img, "cls", noise=noise, return_latents=True)
    e = dict(z0=rel_l2(z0.cpu(), rz0), zt=rel_l2(zt.cpu(), rzt), img=rel_l2(out.cpu(), ref))
    print(f"from-files forward rel-L2 [{dtype}]:", e)
    assert max(e.values()) < tol, e
    assert all(torch.equal(a.cpu(), b) for (_, a), (_, b) in zip(sorted(p.state_dict().items()), sorted(o.state_dict().items())))
