"""Which torch-owned kernels run inside one forward (eager pass under the torch profiler)?   python tools/torch_glue.py"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import build_model

m = build_model(20, "cuda", 0, 1)
m.use_graph = False
img = torch.rand(8, 3, 512, 512, device="cuda")
noise = (torch.randn(8, 4, 64, 64, device="cuda"), torch.randn(8, 4, 64, 64, device="cuda"))
m(img, "ir", noise=noise); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    m(img, "ir", noise=noise); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in prof.key_averages()]
ops_ = [(k, c, t) for k, c, t in rows if k.startswith("aten::") and t > 0]
for k, c, t in sorted(ops_, key=lambda r: -r[2])[:25]:
    print(f"{k:40s} n={c:5d}  {t / 1e3:8.2f} ms")
