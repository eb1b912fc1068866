"""Functional check of the BASELINE.json configurations at full model size on one GPU (random weights)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, bench
dev = torch.device("cuda", 0)
def run(name, steps, shape, task, b):
    m = bench.build_model(steps, dev, 0, 1)
    g = torch.Generator(device=dev).manual_seed(1)
    img = torch.rand(b, 3, *shape, generator=g, device=dev)
    t0 = time.time(); y = m(img, task); torch.cuda.synchronize(); t1 = time.time()
    y = m(img, task); torch.cuda.synchronize(); t2 = time.time()
    print(f"{name}: out {tuple(y.shape)} finite={bool(torch.isfinite(y).all())} range=[{float(y.min()):.3f},{float(y.max()):.3f}] "
          f"first {t1-t0:.2f}s replay {(t2-t1)*1e3:.1f} ms  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    del m; torch.cuda.empty_cache()
run("cfg1 256x256 B=1 4 steps (upscaled to 512)", 4, (256, 256), "ir", 1)
run("cfg1b 300x500 B=1 1 step (resize+pad)", 1, (300, 500), "cls", 1)
run("cfg4 1024x1024 B=1 20 steps seg", 20, (1024, 1024), "seg", 1)
run("cfg5 512x512 B=8 50 steps", 50, (512, 512), "ir", 8)
