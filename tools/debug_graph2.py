import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn.functional as F
from unirestore_amd import ops

def graph_of(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = fn()
    return g, out

def poison():
    a = torch.rand(1, 3, 614, 512, device="cuda")
    return F.pad(a, (0, 0, 0, 26), mode="reflect")

x = torch.randn(2, 32, 32, 64, device="cuda").to(torch.bfloat16)
w = ops.pack_conv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), "cuda")
w1 = ops.pack_conv(torch.randn(256, 64) / 8, torch.randn(256), "cuda")
ga, be = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
xt = torch.randn(256, 256, device="cuda")
tests = {
    "torch_mm": lambda: (xt @ xt).relu() @ xt,
    "conv": lambda: ops.conv(x, w),
    "gn": lambda: ops.group_norm(x, ga, be, 32, 1e-5, True),
    "ln": lambda: ops.layer_norm(x, ga, be, 1e-5),
    "linear_splitk": lambda: ops.linear(x.view(1, 2048, 64)[:, :64].contiguous(), w1),
    "avgpool": lambda: ops.avgpool(x),
    "conv+gn chain": lambda: ops.group_norm(ops.conv(ops.group_norm(ops.conv(x, w), ga, be, 32, 1e-5, True), w), ga, be, 32, 1e-5, True),
}
for name, fn in tests.items():
    g, out = graph_of(fn)
    g.replay(); torch.cuda.synchronize(); r0 = out.clone()
    g.replay(); torch.cuda.synchronize(); r1 = out.clone()
    j = poison(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize(); r2 = out.clone()
    print(name, "replay-replay", float((r0.float() - r1.float()).abs().max()), "after-pad", float((r0.float() - r2.float()).abs().max()))
