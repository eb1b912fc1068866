"""Per-shape HIP-event table of ONE phase (eager, events around every launch): UR_PROF_SHAPES=1 python tools/prof_phase.py encode|decode|controller|step"""
import os, sys
os.environ.setdefault("UR_PROF_SHAPES", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import bench
from unirestore_amd import ops, schedule

which = sys.argv[1] if len(sys.argv) > 1 else "encode"
dev = torch.device("cuda", 0)
m = bench.build_model(20, dev, 0, 1)
m._prepare()
B = 8
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand(B, 3, 512, 512, generator=g, device=dev)
nv, nt = torch.randn(B, 4, 64, 64, generator=g, device=dev), torch.randn(B, 4, 64, 64, generator=g, device=dev)
plan = (512, 512, 0, 0)
with torch.no_grad():
    z0, z0b, mids = m.ae.encode_run(img, nv, enable_fr=True, plan=plan)
    ac = schedule.alphas_cumprod_f64()
    zt, ztb = ops.add_noise(z0, nt, 4, float(np.float32(ac[999] ** 0.5)), float(np.float32((1 - ac[999]) ** 0.5)))
    controls = m.controller.run_schedule(m.controller.stem(z0b), 20)
    fns = {"encode": lambda: m.ae.encode_run(img, nv, enable_fr=True, plan=plan),
           "controller": lambda: m.controller.run_schedule(m.controller.stem(z0b), 20),
           "step": lambda: m.base_model.run(ztb, controls[0], 0),
           "decode": lambda: m.ae.decode_run(zt, mids, "ir", out_plan=((512, 512), (512, 512), False))}
    fn = fns[which]
    fn(); torch.cuda.synchronize()
    ops.profile_enable(True)
    fn(); torch.cuda.synchronize()
    rep = ops.profile_report()
    ops.profile_enable(False)
tot = sum(v["ms"] for v in rep.values())
print(f"{which}: total profiled {tot:.2f} ms, {sum(v['launches'] for v in rep.values())} launches")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("TOP", 45))]:
    tf = v["flops"] / (v["ms"] / 1e3) / 1e12 if v["flops"] else 0.0
    gb = v["bytes"] / (v["ms"] / 1e3) / 1e9 if v["bytes"] else 0.0
    print(f"{k:58s} n={v['launches']:4d} {v['ms']:8.3f} ms  avg {v['ms'] * 1e3 / v['launches']:8.1f} us  {tf:7.1f} TF/s {gb:7.0f} GB/s")
