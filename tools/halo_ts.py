"""Phase timers of the halo conv (A/B build -DUR_HALO_ABL=6): UR_LIB=unirestore_amd/ab/libur_ts.so python tools/halo_ts.py [cin cout]"""
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
t = y.view(-1)[:512].view(torch.int64).cpu().view(-1)[:128]
names = ["issue", "H1", "dma_wait", "barrier M", "H2", "barrier E"]
for w, off in (("wave 0", 0), ("wave NW/2", 64)):
    ts = t[off:off + 63].view(9, 7)
    print(w, "(cycles per phase, taps 0..8 of chunk 1)")
    print("      " + " ".join(f"{n:>10s}" for n in names) + "      total")
    for tap in range(9):
        d = [int(ts[tap, i + 1] - ts[tap, i]) for i in range(6)]
        nxt = int(ts[tap + 1, 0] - ts[tap, 0]) if tap < 8 else sum(d)
        print(f"tap {tap} " + " ".join(f"{v:10d}" for v in d) + f" {nxt:10d}")
