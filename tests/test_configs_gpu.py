"""Every BASELINE.json configuration at FULL model size on the GPU (-m gpu), against the CPU oracle with identical weights.

  configs[0]  256x256 (-> 512x512 inside forward), 4 DDIM steps, B=1
  configs[1]  512x512, 20 steps, B=8  -> its B=1 / 1-step sample against the oracle (the oracle needs ~25 s per step-set on
  configs[2]  = configs[1] per GPU       the box's host cores; 20 steps x 8 images would be an hour), both 16-bit types
  configs[3]  1024x1024, task "seg", B=1 per GPU, 1-step sample (latents 128x128, 16384 tokens)
  configs[4]  512x512, 50 steps, fp16: 50-step trajectory finite + deterministic at full size, 50-step PARITY on the tiny model
  + a 300x500 input (bicubic resize -> 512x853 -> reflect pad -> 512x896 -> un-pad -> resize back) and the config entry point.

Tolerances: bf16 = the emulated 16-bit error budget of the full-size architecture x 1.25 (oracle/emulate.py), fp16 = the north-star 1e-3.
"""
import os
import sys

import pytest
import torch

from golden_util import rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# rel-L2 tolerances (z0, zt, image), full-size random-weight model.
#   bf16: DERIVED from the CPU error budget, not from what the HIP path measured: oracle/emulate.py's "operands + stored
#         activations" row of the full-size architecture x 1.25 (= 7.3e-3 / 5.8e-3 / 5.9e-3) - an implementation that rounds
#         where a 16-bit pipeline must round lands there or below (measured r2-r3: 5.4-5.7e-3 / 3.1-4.1e-3 / 3.9-4.2e-3);
#   fp16: the north-star bar itself, 1e-3 (the budget row x 1.25 would be 9.3e-4 / 7.0e-4 / 7.3e-4; measured 7.3e-4 / 6.3-8.3e-4 /
#         5.2-5.7e-4: the HIP path rounds a few tensors the emulation keeps in fp32, e.g. LayerNorm-folded GEMM operands).
from oracle.emulate import BUDGET_FULLSIZE, BUDGET_MARGIN

TOL = {"bf16": tuple(BUDGET_MARGIN * v for v in BUDGET_FULLSIZE["bf16"]["storage"]), "fp16": (1e-3, 1e-3, 1e-3)}


def _kw(steps):
    return dict(frenc=dict(type="CFRM"), cnet=dict(type="scedit", num_inference_steps=steps),
                tedit=dict(type="TFA", prompt_len=1, task=["ir", "cls", "seg"]))


@pytest.fixture(scope="module")
def full():
    """(oracle on the host, HIP model on cuda:0) - full-size architecture, the same seeded random weights."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle.model import DiffUIE as ODiffUIE
    dev = torch.device("cuda", 0)
    m = bench.build_model(1, dev, 0, 1)
    o = ODiffUIE(**_kw(1)).eval()
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    return o, m


def _oracle(o, img, task, noise, steps):
    from oracle import schedule as osched
    o.num_inference_steps, o.timesteps = steps, osched.ddim_timesteps(steps)
    with torch.no_grad():
        return o(img, task, noise=noise, return_latents=True)


def _check(o, m, img, task, steps, dtypes=("bf16", "fp16"), label="", noise_seed=1234):
    from unirestore_amd.modules import resize_pad_plan
    g = torch.Generator().manual_seed(noise_seed)
    h, w, ph, pw = resize_pad_plan(*img.shape[-2:])
    shp = (img.shape[0], 4, (h + ph) // 8, (w + pw) // 8)
    noise = (torch.randn(shp, generator=g), torch.randn(shp, generator=g))
    ref = _oracle(o, img, task, noise, steps)
    m.set_num_inference_steps(steps)
    errs = {}
    for dt in dtypes:
        m.set_dtype(dt)
        got = m(img, task, noise=noise, return_latents=True)
        e = [rel_l2(a.cpu(), b) for a, b in zip(got, ref)]          # (image, z0, zt)
        print(f"{label} [{dt}] rel-L2 image {e[0]:.2e} z0 {e[1]:.2e} zt {e[2]:.2e}")
        assert got[0].shape == img.shape and bool(torch.isfinite(got[0]).all())
        tz0, tzt, timg = TOL[dt]
        assert e[1] < tz0 and e[2] < tzt and e[0] < timg, (label, dt, e)
        errs[dt] = e
    m.set_dtype("bf16")
    return errs


# fp16 regression bound on zt (the hard bar is 1e-3).  Rounds 2-4 measured 7.9-8.2e-4 against an emulated budget of 5.6e-4; round 5 traced
# the gap (tools/fp16_zt_attrib.py: a dtype-INDEPENDENT 1.6e-3 of eps that vanished with UR_FUSE_LN=0) to a bug, not to rounding: the
# split-K reduce of a LayerNorm-folded QKV GEMM ignored the transposed-V output, so at B <= 2 the 8x8-level self-attention read an
# uninitialised V^T (tests/test_ops_gpu.py::test_layernorm_folded_qkv_writes_v_transposed).  Fixed: 5.14-5.16e-4 over the three draws.
FP16_ZT_REGRESSION = 6.5e-4


# Two of the three draws moved behind UR_EXTRA_DRAWS=1 in round 6 to pay (driver limit: 1 200 s for the whole GPU suite) for the
# two tests below that compare what bench.py actually times: the B=8 batch and the 20-step trajectory.
_DRAWS = [(42, 1234)] + ([(142, 2234), (242, 3234)] if os.environ.get("UR_EXTRA_DRAWS") else [])


@pytest.mark.parametrize("img_seed,noise_seed", _DRAWS)
def test_config1_sample_512_one_step(full, img_seed, noise_seed):
    """The sample bench.py reports as parity_vs_oracle (seeds 42 / 1234): B=1, 512x512, 1 DDIM step (measured bf16 5.4e-3 / 4.0e-3 /
    4.0e-3, fp16 7.3e-4 / 5.2e-4 / 5.2e-4 in round 5) - and two more images / noise draws IN the suite (~40 s of CPU oracle each), so
    that the fp16 margin against the hard 1e-3 is not one sample's luck (round-5 run over the three draws: fp16 zt 5.16e-4 /
    5.16e-4 / 5.14e-4, z0 7.3-7.4e-4, image 5.2e-4; bf16 z0 5.39-5.46e-3, zt 3.93-4.02e-3, image 4.10-4.15e-3)."""
    o, m = full
    img = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(img_seed))
    e = _check(o, m, img, "ir", 1, label=f"configs[1] sample 512x512 / 1 step (seeds {img_seed}, {noise_seed})", noise_seed=noise_seed)
    assert e["fp16"][2] < FP16_ZT_REGRESSION, e["fp16"]


def test_config1_batch8_one_step(full):
    """The bench's BATCH (B=8, 512x512; unifie.py:146-150 is called with whole batches, engine_unifie.py:227-236) against the
    oracle.  Kernel dispatch depends on the batch (tile shapes, split-K, key-split attention, whole-image halo tiles - round 5's
    V^T bug lived in such a branch), so B=1 parity says nothing about it.  Images 1 and 6 of the batch go through the CPU oracle
    (one batched call, ~80 s) under the full-size tolerances TOL; the other six are compared with their own B=1 HIP runs under the
    SAME per-tensor bounds (bf16 z0/zt/image TOL, fp16 8e-4) - not the 1.5e-2 of the independence property test."""
    o, m = full
    g = torch.Generator().manual_seed(61)
    img = torch.rand(8, 3, 512, 512, generator=g)
    noise = (torch.randn(8, 4, 64, 64, generator=g), torch.randn(8, 4, 64, 64, generator=g))
    pick = [1, 6]
    ref = _oracle(o, img[pick], "ir", tuple(n[pick] for n in noise), 1)
    m.set_num_inference_steps(1)
    for dt in ("bf16", "fp16"):
        m.set_dtype(dt)
        got = [t.cpu() for t in m(img, "ir", noise=noise, return_latents=True)]
        assert bool(torch.isfinite(got[0]).all())
        tz0, tzt, timg = TOL[dt]
        for j, b in enumerate(pick):
            e = [rel_l2(a[b:b + 1], r[j:j + 1]) for a, r in zip(got, ref)]
            print(f"configs[1] B=8 / 1 step [{dt}] image {b} vs oracle: image {e[0]:.2e} z0 {e[1]:.2e} zt {e[2]:.2e}")
            assert e[1] < tz0 and e[2] < tzt and e[0] < timg, (dt, b, e)
        lim = TOL["bf16"] if dt == "bf16" else (8e-4, 8e-4, 8e-4)
        worst = [0.0, 0.0, 0.0]
        for b in range(8):
            if b in pick:
                continue
            one = [t.cpu() for t in m(img[b:b + 1], "ir", noise=tuple(n[b:b + 1] for n in noise), return_latents=True)]
            e = [rel_l2(a[b:b + 1], s) for a, s in zip(got, one)]
            worst = [max(w, v) for w, v in zip(worst, e)]
            assert e[1] < lim[0] and e[2] < lim[1] and e[0] < lim[2], (dt, b, e)
        print(f"configs[1] B=8 / 1 step [{dt}] six images vs their own B=1 HIP runs, worst: image {worst[0]:.2e} z0 {worst[1]:.2e} zt {worst[2]:.2e}")
    m.set_dtype("bf16")


# 20-step trajectory bounds = 1.5 x measured (round 6), rel-L2 (image, z0, zt-final)
TRAJ20_TOL = {"bf16": (2.0e-2, 8.0e-3, 2.0e-2), "fp16": (3.0e-3, 1.0e-3, 3.0e-3)}


def test_config1_twenty_steps_b1(full):
    """The bench's STEP COUNT: B=1, 512x512, the full 20-step DDIM trajectory against the oracle (unifie.py:146-150; SURVEY 7 warns of
    the x14.6 amplification of an eps error at t=999), both 16-bit types.  Prints zt's rel-L2 after every step (the per-step growth);
    the product runs eagerly once with DiffUIE.trace_zt collecting the latents, and the graph replay must agree with it bit for bit."""
    o, m = full
    g = torch.Generator().manual_seed(62)
    img = torch.rand(1, 3, 512, 512, generator=g)
    noise = (torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g))
    o.trace_zt = []
    try:
        ref = _oracle(o, img, "ir", noise, 20)
        ref_traj = o.trace_zt
    finally:
        o.trace_zt = None
    assert len(ref_traj) == 20
    m.set_num_inference_steps(20)
    for dt in ("bf16", "fp16"):
        m.set_dtype(dt)
        m.use_graph, m.trace_zt = False, []
        try:
            eager = [t.cpu() for t in m(img, "ir", noise=noise, return_latents=True)]
            traj = m.trace_zt
        finally:
            m.use_graph, m.trace_zt = True, None
        got = [t.cpu() for t in m(img, "ir", noise=noise, return_latents=True)]
        assert all(torch.equal(a, b) for a, b in zip(eager, got))
        growth = [rel_l2(a, b) for a, b in zip(traj, ref_traj)]
        e = [rel_l2(a, b) for a, b in zip(got, ref)]
        print(f"configs[1] B=1 / 20 steps [{dt}] zt rel-L2 per step: " + " ".join(f"{v:.2e}" for v in growth))
        print(f"configs[1] B=1 / 20 steps [{dt}] rel-L2 image {e[0]:.2e} z0 {e[1]:.2e} zt {e[2]:.2e}")
        timg, tz0, tzt = TRAJ20_TOL[dt]
        assert e[0] < timg and e[1] < tz0 and e[2] < tzt, (dt, e)
    m.set_dtype("bf16")


def _scale_residual_writers(sd, frac, factor, seed):
    """Heavy-tail stress: multiply `frac` of the output channels of every layer that WRITES the residual stream (ResnetBlock2D.conv2,
    attention to_out, FeedForward.net.2, Transformer2DModel.proj_out) by `factor` - the shape real SD-2.x weights give their
    'massive activation' channels.  Returns the edited copy."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() >= 2 and v.shape[0] >= 64 and any(
                t in k for t in (".conv2.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight")):
            idx = torch.randperm(v.shape[0], generator=g)[:max(1, int(round(v.shape[0] * frac)))].to(v.device)
            v = v.clone()
            v[idx] *= factor
        out[k] = v
    return out


@pytest.mark.parametrize("factor", [100.0, 30000.0])
def test_heavy_tailed_weights_fp16_is_loud_bf16_is_the_fallback(full, factor):
    """1 % of the residual-stream writers' output channels scaled x100 / x30000 (full-size model, 512x512, 1 step) against the
    oracle with the same edited weights.  fp16 must either still meet the north-star 1e-3 or refuse loudly (FloatingPointError from
    the overflow -> inf -> NaN -> finite check) - never hand back a silently clipped image.  Measured round 5: x100 -> fp16 image
    5.5e-4 / z0 7.2e-4 / zt 5.5e-4 (bf16 4.6e-3 / 5.8e-3 / 3.2e-3, inside its budget x 1.25); x30000 -> fp16 REFUSES, bf16 (fp32
    range, the documented fall-back) stays finite at 2.1e-2 / 2.2e-2 / 5.0e-3 - the 16-bit budget is a statement about well-scaled
    weights, so at x30000 bf16 is only required to stay finite and under 5e-2."""
    o, m = full
    base = {k: v.clone() for k, v in m.state_dict().items()}
    try:
        sd = _scale_residual_writers(base, 0.01, factor, 7)
        m.load_state_dict(sd)
        o.load_state_dict({k: v.cpu() for k, v in sd.items()})
        img = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(52))
        g = torch.Generator().manual_seed(5234)
        noise = (torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g))
        ref = _oracle(o, img, "ir", noise, 1)
        assert bool(torch.isfinite(ref[0]).all())
        m.set_num_inference_steps(1)
        m.set_dtype("bf16")
        got = m(img, "ir", noise=noise, return_latents=True)
        eb = [rel_l2(a.cpu(), b) for a, b in zip(got, ref)]
        m.set_dtype("fp16")
        try:
            got = m(img, "ir", noise=noise, return_latents=True)
            ef = [rel_l2(a.cpu(), b) for a, b in zip(got, ref)]
        except FloatingPointError as exc:
            ef = None
            assert "bf16" in str(exc)
        print(f"heavy tail x{factor:g}: bf16 image/z0/zt {eb[0]:.2e} {eb[1]:.2e} {eb[2]:.2e}; fp16 " +
              ("REFUSED (overflow -> FloatingPointError)" if ef is None else f"{ef[0]:.2e} {ef[1]:.2e} {ef[2]:.2e}"))
        tz0, tzt, timg = TOL["bf16"] if factor <= 100 else (5e-2, 5e-2, 5e-2)
        assert eb[1] < tz0 and eb[2] < tzt and eb[0] < timg, eb
        assert ef is None or max(ef) < 1e-3, ef
        if factor > 1000:
            assert ef is None, "x30000 outlier channels exceed the fp16 range: the forward must refuse, not return"
            # the evaluator's path (runner.py calls forward(quantize=True)): the 8-bit quantise in the output kernel must not turn
            # the NaN image into finite black pixels (fmaxf(NaN, 0) = 0) and so slip past the finite check (round-5 advisor finding)
            with pytest.raises(FloatingPointError):
                m(img, "ir", noise=noise, quantize=True)
        else:
            q = m(img, "ir", noise=noise, quantize=True).cpu()
            assert bool(torch.isfinite(q).all()) and torch.equal(q, (q * 255).round().clamp(0, 255) / 255)
    finally:
        m.set_dtype("bf16")
        m.load_state_dict(base)
        o.load_state_dict({k: v.cpu() for k, v in base.items()})


def test_config0_256_four_steps(full):
    o, m = full
    img = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(43))
    _check(o, m, img, "ir", 4, label="configs[0] 256x256 / 4 steps")


def test_resized_padded_300x500(full):
    o, m = full
    img = torch.rand(1, 3, 300, 500, generator=torch.Generator().manual_seed(44))
    _check(o, m, img, "cls", 1, label="300x500 -> 512x853 -> pad 512x896 / 1 step")


def test_config3_seg_1024_one_step(full):
    o, m = full
    img = torch.rand(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(45))
    _check(o, m, img, "seg", 1, dtypes=("bf16",), label="configs[3] 1024x1024 seg / 1 step")


def test_config1_full_batch_is_bit_deterministic(full):
    """The headline workload itself (B=8, 512x512, 20 steps, bf16): graph replays and an eager run agree bit for bit."""
    _, m = full
    m.set_num_inference_steps(20)
    m.set_dtype("bf16")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(48)
    img = torch.rand(8, 3, 512, 512, generator=g, device=dev)
    nz = (torch.randn(8, 4, 64, 64, generator=g, device=dev), torch.randn(8, 4, 64, 64, generator=g, device=dev))
    runs = [m(img, "ir", noise=nz, return_latents=True) for _ in range(3)]
    m.use_graph = False
    eager = m(img, "ir", noise=nz, return_latents=True)
    m.use_graph = True
    for r in runs[1:] + [eager]:
        assert all(torch.equal(x, y) for x, y in zip(runs[0], r))
    assert bool(torch.isfinite(runs[0][0]).all())


def test_config4_fifty_steps_fp16_full_size(full):
    """50 DDIM steps in fp16 at full size: finite, in range, bit-identical on a second run (graph replay)."""
    _, m = full
    m.set_num_inference_steps(50)
    m.set_dtype("fp16")
    g = torch.Generator().manual_seed(46)
    from unirestore_amd.data import degrade
    hq = torch.rand(3, 3, 512, 512, generator=g)
    lq = torch.stack([degrade(hq[i], k, g) for i, k in enumerate(("noise", "haze", "lowlight"))])     # the mixed-degradation batch
    nz = (torch.randn(3, 4, 64, 64, generator=g), torch.randn(3, 4, 64, 64, generator=g))
    a = m(lq, "ir", noise=nz, return_latents=True)
    b = m(lq, "ir", noise=nz, return_latents=True)
    assert all(bool(torch.isfinite(t).all()) for t in a) and all(torch.equal(x, y) for x, y in zip(a, b))
    assert float(a[0].abs().max()) < 50.0                     # a diverged trajectory would blow up long before 50 steps
    m.set_dtype("bf16")


def test_config4_fifty_steps_parity_tiny():
    """50-step DDIM trajectory against the oracle (tiny configuration, where the oracle takes ~90 s), both 16-bit types."""
    from oracle.model import DiffUIE as ODiffUIE
    from tiny_cfg import TINY, model_kwargs, randomise_
    import unirestore_amd.modules as M
    o = randomise_(ODiffUIE(**model_kwargs(50), **TINY).eval(), 11)
    g = torch.Generator().manual_seed(47)
    img = torch.rand(2, 3, 64, 64, generator=g)
    nz = (torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 4, 64, 64, generator=g))
    oy = o(img, "ir", noise=nz, return_latents=True)
    for dtype, lim in (("bf16", 7.6e-3), ("fp16", 1.2e-3)):   # 1.5 x measured (bf16 5.0e-3 on z0, fp16 8.0e-4)
        p = M.DiffUIE(**model_kwargs(50), **TINY, dtype=dtype).eval()
        p.load_state_dict(o.state_dict())
        assert p.timesteps.tolist() == list(range(999, 0, -20))
        py = p(img, "ir", noise=nz, return_latents=True)
        e = [rel_l2(a.cpu(), b) for a, b in zip(py, oy)]
        print(f"50-step tiny [{dtype}] rel-L2 image {e[0]:.2e} z0 {e[1]:.2e} zt {e[2]:.2e}")
        assert max(e) < lim, (dtype, e)


def test_cli_validate_config0():
    """configs/val_pir_256_4step.yaml through the config entry point (model build, crop, restore, 8-bit quantise, metrics)."""
    from unirestore_amd import cli
    cfg = cli.load_config(os.path.join(ROOT, "configs", "val_pir_256_4step.yaml"))
    res = cli.validate(cfg, max_batches=3)
    assert res["output_finite"] and res["images"] == 3 and res["dtype"] == "bf16" and res["denoise_steps"] == 4
    assert res["images_per_s"] and res["images_per_s"] > 1.0 and 0.0 < res["val_lq/ssim"] <= 1.0
