import sys, time; sys.path.insert(0, ".")
import torch
from unirestore_amd import ops
B, t, heads, d = 8, 4096, 5, 64
c = heads * d
qkv = torch.randn(B, t, 3 * c, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, c, t, device="cuda").to(torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(200):
        ops.attention(qkv, qkv[:, :, c:], vt, heads, d, t, t, 0.125, ldq=3 * c, ldk=3 * c, bs_q=t * 3 * c, bs_k=t * 3 * c, bs_vt=c * t, batch=B)
    torch.cuda.synchronize()
