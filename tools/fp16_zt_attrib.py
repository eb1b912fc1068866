"""Which part of the UNet step puts fp16 zt above the emulated budget?  B=1, 512x512, 1 DDIM step, full-size weights: the UNet step is
run from the ORACLE's own z0 / zt / control inputs (so encoder error is excluded) under the environment switches given on the command line
of the calling shell (UR_CHAIN=0, UR_FUSE_LN=0, UR_ATTN_NOPP=1 ...), and eps is compared with the oracle's.  The oracle tensors are cached
in gpurun_out/fp16_attrib_ref.pt by the first run.   python tools/fp16_zt_attrib.py <label>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, bench
from unirestore_amd import ops

label = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda", 0)
m = bench.build_model(1, dev, 0, 1)
cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache", "fp16_attrib_ref.pt")     # (tools/_cache travels with gpurun; git-ignored)
if not os.path.exists(cache):
    from oracle.model import DiffUIE as ODiffUIE
    from oracle import schedule as osched
    kw = dict(frenc=dict(type="CFRM"), cnet=dict(type="scedit", num_inference_steps=1), tedit=dict(type="TFA", prompt_len=1, task=["ir", "cls", "seg"]))
    o = ODiffUIE(**kw).eval()
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    g = torch.Generator().manual_seed(1234)
    img = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(42))
    noise = (torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g))
    with torch.no_grad():
        z0, mids = o.ae.encode(img, enable_fr=True, noise=noise[0])
        ts = torch.tensor([999])
        zt = osched.add_noise(z0, noise[1], ts)
        ctl = o.controller(z0, ts)
        eps = o.base_model(zt, ctl, ts)
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    torch.save(dict(z0=z0, zt=zt, ctl=ctl, eps=eps), cache)
ref = torch.load(cache)
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
ts = torch.tensor([999])
for dt in ("fp16", "bf16"):
    m.set_dtype(dt)
    m._prepare()
    pe = m.base_model(ref["zt"], ref["ctl"], ts)                  # UNet + SC-Tuner from the oracle's exact inputs
    pc = m.controller(ref["z0"], ts)
    ec = max(rel(pc[k], ref["ctl"][k]) for k in pc)
    pe2 = m.base_model(ref["zt"], {k: v.float().cpu() for k, v in pc.items()}, ts)      # ... from the HIP Controller's outputs
    print(f"[{label}] {dt}: eps rel-L2 (oracle controls) {rel(pe, ref['eps']):.3e}   controller outputs {ec:.3e}   eps (HIP controls) {rel(pe2, ref['eps']):.3e}")
