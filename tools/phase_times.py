"""Where a forward's time goes, by PHASE, each phase captured as its own hipGraph and replayed (in-graph time, B=8, 512x512, 20 steps):
VAE encode + CFRM | Controller (schedule-batched) | one UNet + SC-Tuner + DDIM step (x20) | VAE decode + TFA.   python tools/phase_times.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import bench
from unirestore_amd import ops, schedule

dev = torch.device("cuda", 0)
m = bench.build_model(20, dev, 0, 1)
m._prepare()
B = int(os.environ.get("B", 8))
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand(B, 3, 512, 512, generator=g, device=dev)
nv, nt = torch.randn(B, 4, 64, 64, generator=g, device=dev), torch.randn(B, 4, 64, 64, generator=g, device=dev)
plan = (512, 512, 0, 0)


def timed(name, fn, reps=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:46s} {ms:8.2f} ms")
    return out, ms, gr


with torch.no_grad():
    (z0, z0b, mids), t_enc, _g1 = timed("VAE encode + CFRM", lambda: m.ae.encode_run(img, nv, enable_fr=True, plan=plan))
    (_, t_enc0, _g1b) = timed("  (VAE encode without CFRM)", lambda: m.ae.encode_run(img, nv, enable_fr=False, plan=plan))
    ac = schedule.alphas_cumprod_f64()
    zt, ztb = ops.add_noise(z0, nt, 4, float(np.float32(ac[999] ** 0.5)), float(np.float32((1 - ac[999]) ** 0.5)))
    (controls, t_ctl, _g2) = timed("Controller, 20 steps in one batched pass", lambda: m.controller.run_schedule(m.controller.stem(z0b), 20))

    def step():
        eps = m.base_model.run(ztb, controls[0], 0)
        c_x, c_e = schedule.ddim_coefficients(int(m.timesteps[0]), 20)
        ops.ddim_step_(zt, ztb, eps, 4, c_x, c_e)
        return eps
    (_, t_step, _g3) = timed("one UNet + SC-Tuner + DDIM step", step)
    (_, t_dec, _g4) = timed("VAE decode + TFA", lambda: m.ae.decode_run(zt, mids, "ir", out_plan=((512, 512), (512, 512), False)))
    (_, t_dec0, _g5) = timed("  (decode, same graph, second measurement)", lambda: m.ae.decode_run(zt, mids, "ir", out_plan=((512, 512), (512, 512), False)))
tot = t_enc + t_ctl + 20 * t_step + t_dec
print(f"sum = {t_enc:.1f} + {t_ctl:.1f} + 20 x {t_step:.2f} + {t_dec:.1f} = {tot:.1f} ms  (CFRM = {t_enc - t_enc0:.1f} ms of the encode)")
