"""Per-tile time line of the persistent halo conv (A/B build -DUR_HALO_ABL=7): UR_LIB=unirestore_amd/ab/libur_tl.so python tools/r6/pws_timeline.py cin cout hw"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from unirestore_amd import ops
cin, cout, hw = (int(a) for a in sys.argv[1:4])
x = torch.randn(8, hw, hw, cin, device="cuda").to(torch.bfloat16)
pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.randn(cout), "cuda")
ws = ops.workspace(x.device)
for _ in range(3):
    y = ops.conv(x, pc)
torch.cuda.synchronize()
ws[(32 << 20):(32 << 20) + 64 * 40 * 8].zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = ops.conv(x, pc); e1.record(); torch.cuda.synchronize()
t = ws[(32 << 20):(32 << 20) + 64 * 40 * 8].view(torch.int64).cpu().view(64, 40, 4).double() / 100.0
print(f"c3 {cin}->{cout}@{hw} B=8: host events {e0.elapsed_time(e1) * 1e3:.1f} us (eager)")
for wg in (0, 1, 63):
    tt = t[wg]; n = int((tt[:, 0] > 0).sum())
    if n == 0:
        print(f"workgroup {wg}: no stamps (not the persistent kernel?)"); continue
    tt = tt[:n]; t0 = tt[0, 0]
    k = tt[:, 1] - tt[:, 0]; e = tt[:, 2] - tt[:, 1]; gap = tt[1:, 0] - tt[:-1, 2] if n > 1 else torch.zeros(1, dtype=torch.double)
    print(f"workgroup {wg}: {n} tiles, total {tt[-1, 2] - t0:.1f} us; per tile: K loop mean {k.mean():.2f} (first {k[0]:.2f}, min {k.min():.2f}, max {k.max():.2f}), "
          f"epilogue mean {e.mean():.2f} (min {e.min():.2f}, max {e.max():.2f}), tile-to-tile gap mean {gap.mean():.2f} us")
