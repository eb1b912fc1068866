"""Workgroup time line of the halo conv (A/B build -DUR_HALO_ABL=7): UR_LIB=unirestore_amd/ab/libur_tl.so python tools/halo_wg_timeline.py [cin cout [hw]]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unirestore_amd import ops
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 320)
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 64
x = torch.randn(8, hw, hw, cin, device="cuda").to(torch.bfloat16)
pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.randn(cout), "cuda")
for _ in range(3):
    y = ops.conv(x, pc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = ops.conv(x, pc); e1.record(); torch.cuda.synchronize()
t = y.view(-1)[:4096].view(torch.int64).cpu().view(256, 4).double() / 100.0     # us
t0 = t[:, 0].min()
print(f"host events: {e0.elapsed_time(e1) * 1e3:.1f} us (eager launch)")
print(f"workgroup start   : first {0.0:.2f}  last {t[:, 0].max() - t0:.2f} us")
print(f"prologue (start -> K loop): mean {(t[:, 1] - t[:, 0]).mean():.2f}  max {(t[:, 1] - t[:, 0]).max():.2f} us")
print(f"K loop            : mean {(t[:, 2] - t[:, 1]).mean():.2f}  min {(t[:, 2] - t[:, 1]).min():.2f}  max {(t[:, 2] - t[:, 1]).max():.2f} us")
print(f"epilogue          : mean {(t[:, 3] - t[:, 2]).mean():.2f}  max {(t[:, 3] - t[:, 2]).max():.2f} us")
print(f"first start -> last end: {t[:, 3].max() - t0:.2f} us;  first K-loop start {t[:, 1].min() - t0:.2f}, last K-loop end {t[:, 2].max() - t0:.2f}")
