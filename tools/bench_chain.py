"""In-graph timing of the fused chains vs the per-layer launches they replace (B=8, 64x64 level: 32768 tokens x 320):
python tools/bench_chain.py        UR_CHAIN_DBG=1 (no MFMA phases) / 2 (no DMA) are timing-only ablations of the chain kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import math, torch
from unirestore_amd import ops, chain
from unirestore_amd.modules import nn as N
from ab_micro import gtime

torch.manual_seed(0)
T, C, H = 32768, 320, 1280
for dtn in ("bf16", "fp16"):
    dt = ops.set_dtype(dtn)
    x = (torch.randn(T, C, device="cuda") * 1.5).to(dt)
    ff = N.FeedForward(C)
    norm = N.LayerNorm(C)
    for p in list(ff.parameters()) + list(norm.parameters()):
        torch.nn.init.normal_(p, std=0.05)
    torch.nn.init.ones_(norm.weight)
    st = chain.pack_mlp(ff.net[0].proj.weight, ff.net[0].proj.bias, ff.net[2].weight, ff.net[2].bias, norm.weight, norm.bias, "cuda")
    y = chain.ff_geglu_fused(x, st, H, 1e-5)
    # the per-layer path: row sums from a producer GEMM are needed for the LN fold -> emulate with an identity-ish producer
    xs = ops.linear(x, ops.pack_conv(torch.eye(C), None, "cuda"), rows=True)
    def per_layer():
        return ff.run(xs, xs, ln=(norm, ops.ln_of(xs)))
    y0 = per_layer()
    d = (y.float() - y0.float()).norm() / y0.float().norm()
    t_f = gtime(lambda: chain.ff_geglu_fused(x, st, H, 1e-5))
    t_p = gtime(per_layer)
    fl = 2.0 * T * C * H * 3
    print(f"[{dtn}] MLP 32768x320 (hidden 1280): fused {t_f:7.1f} us ({fl / t_f / 1e6:6.0f} TF/s, weights {st.numel() * (T // 128) / t_f / 1e3:6.0f} GB/s L2->LDS)"
          f"   per-layer (2 launches) {t_p:7.1f} us   rel diff fused vs per-layer {d:.2e}")

if os.environ.get("UR_CHAIN_STAMPS"):        # an ablation-6 build: cycle stamps of FF chunk 10 of workgroup 17, per wave
    dt = ops.set_dtype("bf16")
    x = (torch.randn(T, C, device="cuda") * 1.5).to(dt)
    y = chain.ff_geglu_fused(x, st if st.dtype == torch.uint8 else st, H, 1e-5)
    torch.cuda.synchronize()
    raw = y.view(torch.int16).reshape(-1)[(T - 128) * C:].view(torch.int64)[:64].cpu().reshape(4, 16)[:, :9]
    names = ["start", "acq1", "ff1 a", "epi a", "acq2", "ff1 b", "epi b", "acq3", "ff2"]
    for w in range(4):
        d = (raw[w][1:] - raw[w][:-1]).tolist()
        print(f"wave {w}: " + "  ".join(f"{n}:{v}" for n, v in zip(names[1:], d)) + f"   chunk total {int(raw[w][8] - raw[w][0])}")
