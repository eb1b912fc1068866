"""Full-size parity sample vs the CPU oracle at an arbitrary input size (default 1024x1024, 1 DDIM step, B=1, task seg)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, bench
from oracle.model import DiffUIE as ODiffUIE
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
resw = int(sys.argv[3]) if len(sys.argv) > 3 else res
task = sys.argv[2] if len(sys.argv) > 2 else "seg"
dev = torch.device("cuda", 0)
m = bench.build_model(1, dev, 0, 1)
kw = dict(frenc=dict(type="CFRM"), cnet=dict(type="scedit", num_inference_steps=1), tedit=dict(type="TFA", prompt_len=1, task=["ir", "cls", "seg"]))
o = ODiffUIE(**kw).eval()
o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
g = torch.Generator().manual_seed(7)
img = torch.rand(1, 3, res, resw, generator=g)
from unirestore_amd.modules import resize_pad_plan
h, w, ph, pw = resize_pad_plan(res, resw)
nz = (torch.randn(1, 4, (h + ph) // 8, (w + pw) // 8, generator=g), torch.randn(1, 4, (h + ph) // 8, (w + pw) // 8, generator=g))
t0 = time.time()
with torch.no_grad():
    oy, oz0, ozt = o(img, task, noise=nz, return_latents=True)
t1 = time.time()
py, pz0, pzt = m(img, task, noise=nz, return_latents=True)
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
print(f"{res}x{resw} task={task}: oracle {t1 - t0:.1f}s on {torch.get_num_threads()} threads;  rel-L2  z0 {rel(pz0, oz0):.3e}  zt {rel(pzt, ozt):.3e}  image {rel(py, oy):.3e}")
