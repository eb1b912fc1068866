"""12 s loop of the dominant conv3x3 launch (320 -> 320 @ 64x64, B = 8) for tools/power_probe.sh."""
import sys, time; sys.path.insert(0, ".")
import torch
from unirestore_amd import ops
x = torch.randn(8, 64, 64, 320, device="cuda").to(torch.bfloat16)
pc = ops.pack_conv(torch.randn(320, 320, 3, 3) / 54, torch.randn(320), "cuda")
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(300):
        ops.conv(x, pc, gn=True)
    torch.cuda.synchronize()
