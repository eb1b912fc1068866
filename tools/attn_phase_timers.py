"""Phase timers of the ping-pong attention kernel (a library built with ATTN_DBG=1 tools/ab_attn.sh "dbg:0:-DUR_ATTN_PP_DBG"):
every wave overwrites the first 16 bytes of its queries' output rows with four cycle sums -
softmax phase (DMA issue + arithmetic) | DMA wait + fragment prefetch + barrier | MFMA phase | barrier behind it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unirestore_amd import ops
B, t, heads, d = 8, 4096, int(os.environ.get("H", "4")), 64
c = heads * d
qkv = torch.randn(B, t, 3 * c, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, c, t, device="cuda").to(torch.bfloat16)
kw = dict(ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=B)
for _ in range(3):
    o = ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, 0.125, **kw)
torch.cuda.synchronize()
raw = o.view(torch.int32).view(B, t, heads, 32)[:, ::32, :, :6].reshape(B, t // 32, heads, 6).double()      # one row per wave
raw = torch.where(raw < 0, raw + 2.0 ** 32, raw)
w = raw[..., :4]
cyc, rt = raw[..., 4].mean().item(), raw[..., 5].mean().item()
print(f"tile loop: {cyc:.0f} shader cycles in {rt:.0f} ticks of 100 MHz = {cyc / rt * 0.1:.3f} GHz; {cyc / (t // 64):.1f} cycles per tile")
nt = t // 64
names = ["softmax phase", "wait+prefetch+barrier", "MFMA phase", "barrier"]
for g, nm in ((0, "group A (waves 0-3)"), (1, "group B (waves 4-7)")):
    sel = w[:, :, :, :].reshape(B, t // 256, 8, heads, 4)[:, :, 4 * g:4 * g + 4].reshape(-1, 4)
    per_tile = sel.mean(0) / nt
    print(nm, " | ".join(f"{n}: {x:7.1f}" for n, x in zip(names, per_tile.tolist())), f"| sum {per_tile.sum():7.1f} cycles per tile")
