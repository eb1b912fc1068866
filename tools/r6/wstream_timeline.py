"""Per-wave time line of the weight-streaming 8x8 conv (A/B build with -DUR_WS_TL=1):
UR_LIB=unirestore_amd/ab/libur_wstl.so python tools/r6/wstream_timeline.py [cin cout B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from unirestore_amd import ops
cin, cout, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1280, 1280, 8)
x = torch.randn(B, 8, 8, cin, device="cuda").to(torch.bfloat16)
pcs = [ops.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.randn(cout), "cuda") for _ in range(12)]
ws = ops.workspace(x.device)
for cold in (False, True):
    for _ in range(3):
        y = ops.conv(x, pcs[0])
    if cold:                                 # 12 distinct weight sets (> the 256-MiB Infinity Cache together): this launch streams from HBM
        for pc in pcs[1:]:
            ops.conv(x, pc)
    torch.cuda.synchronize()
    ws[(32 << 20):(32 << 20) + 65536].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = ops.conv(x, pcs[0]); e1.record(); torch.cuda.synchronize()
    t = ws[(32 << 20):(32 << 20) + 2 * 8 * 4 * 256].view(torch.int64).cpu().view(-1, 8).double() / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    seg = lambda a, b: f"mean {(t[:, b] - t[:, a]).mean():6.2f}  min {(t[:, b] - t[:, a]).min():6.2f}  max {(t[:, b] - t[:, a]).max():6.2f}"
    print(f"--- {'cold (weights from HBM)' if cold else 'warm (weights in the Infinity Cache)'}: {len(t)} waves, host events {e0.elapsed_time(e1) * 1e3:.1f} us (conv + reduce, eager)")
    print(f"wave start spread : {t[:, 0].max() - t0:.2f} us")
    print(f"prologue (weight prefetch + patch DMA landed): {seg(0, 1)}")
    print(f"K loop (36 units)                            : {seg(1, 2)}")
    print(f"exchange (4 barriers)                        : {seg(2, 3)}")
    print(f"partial-plane store                          : {seg(3, 4)}")
    print(f"first start -> last end: {t[:, 4].max() - t0:.2f} us")
