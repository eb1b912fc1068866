"""Replays the hipGraph of ONE UNet + SC-Tuner + DDIM step (B=8, 512x512) a few times - the target of
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/step_trace.py
and tools/step_trace.py --summarize <kernel_trace.csv> groups the trace by (kernel, grid) into in-graph time per step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
REPS = 4

if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    import csv, collections, re
    rows = list(csv.DictReader(open(sys.argv[2])))
    agg = collections.defaultdict(lambda: [0, 0.0])
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cut = 0
    for i in range(1, len(rows)):                            # the replays follow a 0.5 s sleep: keep what comes after the last long gap
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 200_000_000:
            cut = i
    rows = rows[cut:]
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
    # keep the LAST `REPS` replays only: the kernels of the warm-up / capture runs come first; a replay = identical sequences, so use counts
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        m = re.match(r"([A-Za-z0-9_:]+(<[^(]{0,48}>)?)", n)
        key = ((m.group(1) if m else n)[:70], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    runs = REPS + 1
    tot = sum(v[1] for v in agg.values())
    nk = sum(v[0] for v in agg.values())
    print(f"{runs} replays of the step graph: {nk // runs} kernels per step, sum of kernel durations {tot / 1e3 / runs:.3f} ms per step, wall {span / runs:.3f} ms per step "
          f"(gaps between kernels {(span - tot / 1e3) / runs:.3f} ms)")
    print(f"{'kernel':72s} {'grid':>9s} {'per step':>8s} {'avg us':>8s} {'ms/step':>8s} {'share':>6s}")
    for (k, g), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
        print(f"{k:72s} {g:>9s} {c / runs:8.1f} {us / c:8.2f} {us / 1e3 / runs:8.3f} {100 * us / tot:5.1f}%")
    sys.exit(0)

import numpy as np
import torch, bench
from unirestore_amd import ops, schedule
dev = torch.device("cuda", 0)
m = bench.build_model(20, dev, 0, 1)
m._prepare()
B = 8
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand(B, 3, 512, 512, generator=g, device=dev)
nv, nt = torch.randn(B, 4, 64, 64, generator=g, device=dev), torch.randn(B, 4, 64, 64, generator=g, device=dev)
with torch.no_grad():
    z0, z0b, mids = m.ae.encode_run(img, nv, enable_fr=True, plan=(512, 512, 0, 0))
    ac = schedule.alphas_cumprod_f64()
    zt, ztb = ops.add_noise(z0, nt, 4, float(np.float32(ac[999] ** 0.5)), float(np.float32((1 - ac[999]) ** 0.5)))
    controls = m.controller.run_schedule(m.controller.stem(z0b), 20)

    def step():
        eps = m.base_model.run(ztb, controls[0], 0)
        c_x, c_e = schedule.ddim_coefficients(int(m.timesteps[0]), 20)
        ops.ddim_step_(zt, ztb, eps, 4, c_x, c_e)
    step(); torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("MARK begin step phase", flush=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    gr.replay()
    torch.cuda.synchronize()
    import time
    time.sleep(0.5)
    for _ in range(REPS + 1):
        gr.replay()
    torch.cuda.synchronize()
