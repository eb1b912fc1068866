"""Per-shape timing of the igemm kernel on the real layer shapes (B images, 512x512 input -> 64x64 latent)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from unirestore_amd import ops


def gtime(f, reps=20):
    """GPU time per call: `reps` launches captured in one hipGraph (no host launch overhead), replayed 3x."""
    for _ in range(2): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


B = int(os.environ.get("B", 8))
# (name, H, W, Cin, Cout, k, stride, count per forward, act)
S = 20
shapes = [
    # UNet conv3x3 (per step counts x20)
    ("unet c3 320->320@64", 64, 64, 320, 320, 3, 1, 7 * S), ("unet c3 640->320@64", 64, 64, 640, 320, 3, 1, 2 * S),
    ("unet c3 960->320@64", 64, 64, 960, 320, 3, 1, 1 * S), ("unet c3 640->640@64up", 64, 64, 640, 640, 3, 1, 1 * S),
    ("unet c3 320->640@32", 32, 32, 320, 640, 3, 1, 1 * S), ("unet c3 640->640@32", 32, 32, 640, 640, 3, 1, 6 * S),
    ("unet c3 1280->640@32", 32, 32, 1280, 640, 3, 1, 1 * S), ("unet c3 1920->640@32", 32, 32, 1920, 640, 3, 1, 1 * S),
    ("unet c3 1280->1280@32up", 32, 32, 1280, 1280, 3, 1, 1 * S),
    ("unet c3 640->1280@16", 16, 16, 640, 1280, 3, 1, 1 * S), ("unet c3 1280->1280@16", 16, 16, 1280, 1280, 3, 1, 7 * S),
    ("unet c3 2560->1280@16", 16, 16, 2560, 1280, 3, 1, 2 * S), ("unet c3 1280->1280@8", 8, 8, 1280, 1280, 3, 1, 11 * S),
    ("unet c3 2560->1280@8", 8, 8, 2560, 1280, 3, 1, 3 * S),
    # UNet transformer GEMMs (5 blocks per level)
    ("xf qkv 320->960 T4096", 1, 4096, 320, 960, 1, 1, 5 * S), ("xf out/proj 320->320 T4096", 1, 4096, 320, 320, 1, 1, 25 * S),
    ("xf geglu 320->2560 T4096", 1, 4096, 320, 2560, 1, 1, 5 * S), ("xf ff2 1280->320 T4096", 1, 4096, 1280, 320, 1, 1, 5 * S),
    ("xf qkv 640->1920 T1024", 1, 1024, 640, 1920, 1, 1, 5 * S), ("xf out/proj 640->640 T1024", 1, 1024, 640, 640, 1, 1, 25 * S),
    ("xf geglu 640->5120 T1024", 1, 1024, 640, 5120, 1, 1, 5 * S), ("xf ff2 2560->640 T1024", 1, 1024, 2560, 640, 1, 1, 5 * S),
    ("xf qkv 1280->3840 T256", 1, 256, 1280, 3840, 1, 1, 5 * S), ("xf out/proj 1280->1280 T256", 1, 256, 1280, 1280, 1, 1, 25 * S),
    ("xf geglu 1280->10240 T256", 1, 256, 1280, 10240, 1, 1, 5 * S), ("xf ff2 5120->1280 T256", 1, 256, 5120, 1280, 1, 1, 5 * S),
    ("xf out/proj 1280->1280 T64", 1, 64, 1280, 1280, 1, 1, 5 * S),
    # SC-Tuner 1x1
    ("csce 256->320@64", 64, 64, 256, 320, 1, 1, 3 * S), ("csce 320->320@64", 64, 64, 320, 320, 1, 1, 6 * S),
    ("csce 640->640@32", 32, 32, 640, 640, 1, 1, 4 * S), ("csce 1280->1280@16", 16, 16, 1280, 1280, 1, 1, 4 * S),
    ("csce 1280->1280@8", 8, 8, 1280, 1280, 1, 1, 6 * S),
    # Controller
    ("ctrl c3 256->256@64", 64, 64, 256, 256, 3, 1, 6 * S), ("ctrl c3 256->256@32", 32, 32, 256, 256, 3, 1, 6 * S),
    ("ctrl c3 512->512@16", 16, 16, 512, 512, 3, 1, 3 * S), ("ctrl c3 512->512@8", 8, 8, 512, 512, 3, 1, 8 * S),
    # VAE (once)
    ("vae c3 128->128@512", 512, 512, 128, 128, 3, 1, 10), ("vae c3 256->256@256", 256, 256, 256, 256, 3, 1, 8),
    ("vae c3 512->512@128", 128, 128, 512, 512, 3, 1, 10), ("vae c3 512->512@64", 64, 64, 512, 512, 3, 1, 14),
    ("vae c3 256->256@512up", 256, 256, 256, 256, 3, 1, 1), ("vae c3 128->8@512", 512, 512, 128, 8, 3, 1, 1),
    # round 6: the slow half of the conv3x3 family (stride 2, the 8x8 -> 16x16 upsampling conv, the schedule-batched Controller)
    ("s2 c3 320->320@64", 64, 64, 320, 320, 3, 2, 1 * S), ("s2 c3 640->640@32", 32, 32, 640, 640, 3, 2, 1 * S),
    ("s2 c3 1280->1280@16", 16, 16, 1280, 1280, 3, 2, 1 * S), ("unet c3 1280->1280@16up", 8, 8, 1280, 1280, 3, 1, 1 * S),
    ("ctrlB c3 256->256@64 x20", 64, 64, 256, 256, 3, 1, 6, 20), ("ctrlB c3 256->256@32 x20", 32, 32, 256, 256, 3, 1, 6, 20),
    ("ctrlB c3 512->512@16 x20", 16, 16, 512, 512, 3, 1, 3, 20), ("ctrlB c3 512->512@8 x20", 8, 8, 512, 512, 3, 1, 8, 20),
]
only = os.environ.get("ONLY")
tot = 0.0
print(f"{'shape':32s} {'M':>8s} {'us':>9s} {'TF/s':>7s} {'cnt':>5s} {'ms/fwd':>8s}")
REPS = int(os.environ.get("REPS", 20))
for name, h, w, cin, cout, k, stride, cnt, *bm in shapes:
    if only and only not in name: continue
    up = name.endswith("up")
    Bx = B * (bm[0] if bm else 1)
    x = torch.randn(Bx, h, w, cin, device="cuda").to(torch.bfloat16)
    pc = ops.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.randn(cout), "cuda")
    f = lambda: ops.conv(x, pc, upsample=up, stride=stride)       # stride 2: the UNet's Downsample2D (padding 1)
    us = gtime(f, REPS)
    m = Bx * h * w * (4 if up else 1) // (stride * stride)
    fl = 2.0 * m * cout * cin * k * k
    tot += us * cnt / 1e3
    print(f"{name:32s} {m:8d} {us:9.1f} {fl / us / 1e6:7.1f} {cnt:5d} {us * cnt / 1e3:8.2f}")
print("total ms/forward (listed shapes):", round(tot, 1))
