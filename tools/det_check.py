"""Bit-determinism probe at full size: python tools/det_check.py [steps] [batch] [dtype]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
steps, b, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda", 0)
m = bench.build_model(steps, dev, 0, 1, dt)
g = torch.Generator(device=dev).manual_seed(3)
img = torch.rand(b, 3, 512, 512, generator=g, device=dev)
nz = (torch.randn(b, 4, 64, 64, generator=g, device=dev), torch.randn(b, 4, 64, 64, generator=g, device=dev))
outs = [m(img, "ir", noise=nz, return_latents=True) for _ in range(4)]
for i in range(1, 4):
    print(f"steps={steps} B={b} {dt} replay0 vs replay{i}:",
          [bool(torch.equal(x, y)) for x, y in zip(outs[0], outs[i])], [float((x - y).abs().max()) for x, y in zip(outs[0], outs[i])])
m.use_graph = False
e = [m(img, "ir", noise=nz, return_latents=True) for _ in range(2)]
print("eager vs eager:", [bool(torch.equal(x, y)) for x, y in zip(e[0], e[1])], " eager vs graph:", [bool(torch.equal(x, y)) for x, y in zip(e[0], outs[0])])
