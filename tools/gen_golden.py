"""Generate tests/golden/*.npz by importing the reference's own adapter classes.

Runs ONLY in the build container (needs /root/reference).  The reference Python never travels:
only the produced vectors (inputs, weights, outputs; fp32) are committed.  Two shims make
nafnet_arch.py / cfrm.py importable without timm / diffusers (neither is installed):
  timm.layers.LayerNorm2d  -> LayerNorm over C of an NCHW tensor, eps 1e-6 (timm's definition)
  diffusers.AutoencoderKL  -> dummy symbol (imported by cfrm.py, unused by the classes we call)
Zero-initialised parameters (NAFBlock beta/gamma) are re-randomised so the fixtures are not vacuous.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/src/modules/diffuie"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _shims():
    class LayerNorm2d(nn.LayerNorm):
        def __init__(self, c, eps=1e-6):
            super().__init__(c, eps=eps)

        def forward(self, x):
            return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)

    timm = types.ModuleType("timm"); layers = types.ModuleType("timm.layers")
    layers.LayerNorm2d = LayerNorm2d; timm.layers = layers
    sys.modules.setdefault("timm", timm); sys.modules.setdefault("timm.layers", layers)
    diff = types.ModuleType("diffusers"); diff.AutoencoderKL = object
    sys.modules.setdefault("diffusers", diff)


def _load(name, pkg="refdiffuie"):
    if pkg not in sys.modules:
        p = types.ModuleType(pkg); p.__path__ = [REF]; sys.modules[pkg] = p
    spec = importlib.util.spec_from_file_location(f"{pkg}.{name}", os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec); sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def _randomise(m, gen):
    for p in m.parameters():
        if p.dim() > 1 and float(p.abs().sum()) == 0.0 or (p.dim() == 4 and p.shape[0] == 1 and float(p.abs().sum()) == 0):
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=gen) * 0.5)


def _save(name, module, inputs, outputs):
    if os.path.exists(os.path.join(OUT, name + ".npz")) and "--force" not in sys.argv:
        print(name, "exists (kept; --force regenerates)")
        return
    d = {f"w::{k}": v.detach().numpy() for k, v in module.state_dict().items()}
    d.update({f"in::{k}": v.numpy() for k, v in inputs.items()})
    d.update({f"out::{k}": v.detach().numpy() for k, v in outputs.items() if v is not None})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, {k: tuple(v.shape) for k, v in outputs.items() if v is not None})


@torch.no_grad()
def main():
    _shims()
    scedit, taskeditor, spade = _load("scedit"), _load("taskeditor"), _load("spade")
    naf = _load("nafnet_arch"); cfrm = _load("cfrm")
    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(20250614)
    rn = lambda *s: torch.randn(*s, generator=g)

    for i, (c, cc, h, w) in enumerate([(32, 16, 8, 8), (64, 32, 12, 20)]):
        torch.manual_seed(100 + i)
        m = scedit.CSCEAdapter(c, c, cc).eval()
        x, cond = rn(2, c, h, w), rn(2, cc, h, w)
        _save(f"csce_{i}", m, dict(x=x, condition=cond), dict(y=m(x, cond)))

    for i, (co, cs, t, last, h, w) in enumerate([(64, 64, 1, False, 8, 8), (64, 32, 1, False, 10, 12),
                                                  (64, 16, 1, True, 16, 8), (64, 32, 2, False, 8, 8)]):
        torch.manual_seed(200 + i)
        m = taskeditor.TaskFeatureAdapter(co, cs, t, last).eval()
        x, skip, cond = rn(2, co, h, w), rn(2, cs, h, w), rn(2, t, cs)
        y, nc = m(x, skip, cond)
        _save(f"tfa_{i}", m, dict(x=x, skip=skip, condition=cond), dict(x=y, condition=nc))

    for i, (c, n, h, w) in enumerate([(16, 1, 9, 7), (32, 2, 8, 12)]):
        torch.manual_seed(300 + i)
        m = nn.Sequential(*[naf.NAFBlock(c) for _ in range(n)], cfrm.AdaNAFV2(c)).eval()
        _randomise(m, g)
        x = rn(2, c, h, w)
        _save(f"cfrm_{i}", m, dict(x=x), dict(y=m(x)))

    torch.manual_seed(400)
    m = spade.SPADE(64, 32).eval()
    x, seg = rn(2, 64, 16, 16), rn(2, 32, 8, 8)
    _save("spade_0", m, dict(x=x, segmap=seg), dict(y=m(x, seg)))

    # production-width SC-Tuner ((C, Cc, h, w) = (320, 256, 8, 8)); its own generator so that the vectors above
    # do not move when cases are appended
    g2 = torch.Generator().manual_seed(20250615)
    torch.manual_seed(102)
    m = scedit.CSCEAdapter(320, 320, 256).eval()
    x, cond = torch.randn(2, 320, 8, 8, generator=g2), torch.randn(2, 256, 8, 8, generator=g2)
    _save("csce_2", m, dict(x=x, condition=cond), dict(y=m(x, cond)))

    # Round 4: shapes that reach the kernels PRODUCTION runs, each from its own generator.
    #  csce_3: CSCEAdapter(320,320,256) at 16x16 = 256 tokens per image -> the token-stationary tchain_csce_kernel (scedit.py:24-38)
    #  csce_4: CSCEAdapter(640,640,256) at 8x16 -> the 640-wide per-layer path with the schedule-batched proj(cond)
    #  cfrm_2: Sequential(NAFBlock(128), AdaNAFV2(128)) at 16x16 -> the 128/256/512-wide CFRM tiles (cfrm.py:12-54, nafnet_arch.py:28-131)
    #  tfa_4 : TaskFeatureAdapter(512,128,1,last_layer=True) at 16x16, 1.26 M parameters (taskeditor.py:10-108)
    g3 = torch.Generator().manual_seed(20250916)
    torch.manual_seed(103)
    m = scedit.CSCEAdapter(320, 320, 256).eval()
    x, cond = torch.randn(2, 320, 16, 16, generator=g3), torch.randn(2, 256, 16, 16, generator=g3)
    _save("csce_3", m, dict(x=x, condition=cond), dict(y=m(x, cond)))
    torch.manual_seed(104)
    m = scedit.CSCEAdapter(640, 640, 256).eval()
    x, cond = torch.randn(2, 640, 8, 16, generator=g3), torch.randn(2, 256, 8, 16, generator=g3)
    _save("csce_4", m, dict(x=x, condition=cond), dict(y=m(x, cond)))
    torch.manual_seed(302)
    m = nn.Sequential(naf.NAFBlock(128), cfrm.AdaNAFV2(128)).eval()
    _randomise(m, g3)
    x = torch.randn(2, 128, 16, 16, generator=g3)
    _save("cfrm_2", m, dict(x=x), dict(y=m(x)))
    torch.manual_seed(204)
    m = taskeditor.TaskFeatureAdapter(512, 128, 1, True).eval()
    x, skip, cond = torch.randn(2, 512, 16, 16, generator=g3), torch.randn(2, 128, 16, 16, generator=g3), torch.randn(2, 1, 128, generator=g3)
    y, nc = m(x, skip, cond)
    assert sum(p.numel() for p in m.parameters()) == 1263232       # SURVEY 8(c) KAT
    _save("tfa_4", m, dict(x=x, skip=skip, condition=cond), dict(x=y, condition=nc))


if __name__ == "__main__":
    main()
