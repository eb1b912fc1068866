import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, torch.nn.functional as F
import unirestore_amd.modules as M
from unirestore_amd import ops
from tiny_cfg import TINY, model_kwargs, randomise_
from golden_util import rel_l2
torch.manual_seed(0)
p = randomise_(M.DiffUIE(**model_kwargs(2), **TINY, use_graph=False).eval(), 3)
p.refresh(); p._prepare()
g = torch.Generator().manual_seed(7)
imgs = [torch.rand(1, 3, 640, 512, generator=g).cuda() for _ in range(2)]
nz = [torch.randn(1, 4, 80, 64, generator=g).cuda() for _ in range(2)]
ref = [[t.clone() for t in p._forward_device(im, "ir", nz[0], nz[1])] for im in imgs]

def persistent():
    out = {}
    def add(name, v):
        if isinstance(v, torch.Tensor) and v.is_cuda: out[name] = v
        elif isinstance(v, (tuple, list)):
            for j, e in enumerate(v): add(f"{name}.{j}", e)
        elif isinstance(v, ops.PackedConv): add(name + ".w", v.w); add(name + ".b", v.bias)
    for mn, m in p.named_modules():
        for k, v in m.__dict__.items():
            if isinstance(k, str) and k.startswith("_"): continue
            add(f"{mn}:{k}", v)
    for k, v in ops._ws.items(): pass
    return out
def sums(d): return {k: float(v.double().abs().sum()) for k, v in d.items()}

out = p._graph_forward(imgs[0], "ir", nz[0], nz[1]); torch.cuda.synchronize()
print("first", [round(rel_l2(a.cpu(), b.cpu()), 4) for a, b in zip(out, ref[0])])
pers = persistent(); s0 = sums(pers); print(len(pers), "persistent tensors")
gst = p._graphs[list(p._graphs)[0]][1]
print("ptrs static", {k: hex(v.data_ptr()) for k, v in gst.items()})
a = torch.rand(1, 3, 614, 512, device="cuda"); junk = F.pad(a, (0, 0, 0, 26), mode="reflect"); torch.cuda.synchronize()
print("junk ptrs", hex(a.data_ptr()), hex(junk.data_ptr()), junk.numel() * 4)
s1 = sums(pers)
print("changed after pad:", [k for k in s0 if s0[k] != s1[k]])
out = p._graph_forward(imgs[0], "ir", nz[0], nz[1]); torch.cuda.synchronize()
print("second", [round(rel_l2(a.cpu(), b.cpu()), 4) for a, b in zip(out, ref[0])])
s2 = sums(pers)
print("changed after replay:", [k for k in s0 if s0[k] != s2[k]])
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
