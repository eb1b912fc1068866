"""Phase timers of the halo conv (A/B build -DUR_HALO_ABL=6): UR_LIB=unirestore_amd/ab/libur_ts.so python tools/halo_ts.py [cin cout]
(UR_HALO_NOWS=1: the kernel whose waves load for themselves)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unirestore_amd import ops
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 320)
x = torch.randn(8, 64, 64, cin, device="cuda").to(torch.bfloat16)
pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.randn(cout), "cuda")
for _ in range(3):
    y = ops.conv(x, pc)
torch.cuda.synchronize()
t = y.view(-1)[:1024].view(torch.int64).cpu().view(-1)[:256]
lp = y.view(-1)[1024:1024 + 2048].view(torch.int64).cpu().view(256, 2).double()
if not os.environ.get("UR_HALO_NOWS"):
    print(f"K loop over 256 workgroups: {lp[:, 0].mean():.0f} shader cycles (min {lp[:, 0].min():.0f} max {lp[:, 0].max():.0f}), "
          f"{lp[:, 1].mean() / 100:.2f} us, clock {lp[:, 0].mean() / (lp[:, 1].mean() * 10):.3f} GHz")
ws = not os.environ.get("UR_HALO_NOWS")
if ws:
    sets = (("compute wave 0", 0, ["H1", "barrier M", "H2", "barrier E"]), ("compute wave NW/2", 64, ["H1", "barrier M", "H2", "barrier E"]),
            ("loader wave", 128, ["issue 1", "vmcnt", "barrier M", "issue 2"]))
    n = 5
else:
    names = ["issue", "H1", "dma_wait", "barrier M", "H2", "barrier E"]
    sets = (("wave 0", 0, names), ("wave NW/2", 64, names))
    n = 7
for w, off, names in sets:
    ts = t[off:off + 9 * n].view(9, n)
    print(w, "(cycles per phase, taps 0..8 of chunk 1)")
    print("      " + " ".join(f"{nm:>10s}" for nm in names) + "      total")
    for tap in range(9):
        d = [int(ts[tap, i + 1] - ts[tap, i]) for i in range(n - 1)]
        nxt = int(ts[tap + 1, 0] - ts[tap, 0]) if tap < 8 else sum(d)
        print(f"tap {tap} " + " ".join(f"{v:10d}" for v in d) + f" {nxt:10d}")
