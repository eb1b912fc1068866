"""Per-shape igemm time of one full-size forward (eager, HIP events): UR_PROF_SHAPES=1 python tools/prof_shapes.py"""
import os, sys, json
os.environ["UR_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, bench
from unirestore_amd import ops
dev = torch.device("cuda", 0)
m = bench.build_model(20, dev, 0, 1); m.use_graph = False
img = torch.rand(8, 3, 512, 512, device=dev); nz = (torch.randn(8, 4, 64, 64, device=dev), torch.randn(8, 4, 64, 64, device=dev))
m(img, "ir", noise=nz); torch.cuda.synchronize()
ops.profile_enable(True); m(img, "ir", noise=nz); torch.cuda.synchronize(); rep = ops.profile_report(); ops.profile_enable(False)
rows = sorted(rep.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in rep.values())
print(f"total profiled {tot:.1f} ms")
for k, v in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    tf = v["flops"] / (v["ms"] / 1e3) / 1e12 if v["flops"] else 0
    print(f"{k:58s} n={v['launches']:5d} {v['ms']:8.2f} ms  avg {v['ms']*1e3/v['launches']:8.1f} us  {tf:7.1f} TF/s")
