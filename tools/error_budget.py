#!/usr/bin/env python
"""Error budget of 16-bit arithmetic on the hot path, measured on the CPU oracle (no GPU needed).

  python tools/error_budget.py [--full] [--size 128] [--steps 1]

For each of bf16 / fp16: rel-L2 of (z0, zt, image) against the fp32 oracle with (a) operands of every contraction
rounded ("operand": the floor for 16-bit matrix-core inputs), (b) additionally every stored activation rounded
("storage": what keeping activations as 16-bit tensors in HBM adds).  --full uses the full-size architecture
(sd-turbo widths), otherwise the tiny test configuration.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from oracle import schedule as osched  # noqa: E402
from oracle.emulate import rel_l2, rounding  # noqa: E402
from oracle.model import DiffUIE  # noqa: E402
from tiny_cfg import TINY, model_kwargs, randomise_  # noqa: E402


def pipeline(o, img, noise, steps):
    z0, mids = o.ae.encode(img, enable_fr=True, noise=noise[0])
    zt = osched.add_noise(z0, noise[1], torch.tensor([999]))
    for t in osched.ddim_timesteps(steps):
        ts = torch.tensor([int(t)])
        eps = o.base_model(zt, o.controller(z0, ts), ts)
        zt = osched.ddim_step(eps, int(t), zt, steps)
    return z0, zt, o.ae.decode(zt, mids, "ir")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    torch.manual_seed(a.seed)
    kw = model_kwargs(a.steps)
    if a.full:
        sys.path.insert(0, ROOT)
        from bench import init_random_
        o = DiffUIE(**kw).eval()
        init_random_(o, 42, "cpu")
    else:
        o = randomise_(DiffUIE(**kw, **TINY).eval(), a.seed)
    g = torch.Generator().manual_seed(1)
    s = a.size
    img = torch.rand(1, 3, s, s, generator=g)
    noise = (torch.randn(1, 4, s // 8, s // 8, generator=g), torch.randn(1, 4, s // 8, s // 8, generator=g))
    with torch.no_grad():
        t0 = time.time()
        ref = pipeline(o, img, noise, a.steps)
        print(f"fp32 oracle: {time.time() - t0:.1f}s")
        for dt in (torch.bfloat16, torch.float16):
            for storage in (False, True):
                with rounding(o, dt, operands=True, storage=storage):
                    out = pipeline(o, img, noise, a.steps)
                e = [rel_l2(x, r) for x, r in zip(out, ref)]
                print(f"{str(dt):16s} {'operand+storage' if storage else 'operand only   '}  z0 {e[0]:.2e}  zt {e[1]:.2e}  image {e[2]:.2e}")


if __name__ == "__main__":
    main()
